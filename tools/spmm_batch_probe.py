"""Batched tile SpMM: time per launch for G panels, bytes per unit as in SURVEY 8d.
python tools/spmm_batch_probe.py [N]   (RICADI_SPMM_PRIVATE=1: per-group value arrays)"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from optconpy_amd import _lib, problems as pb
N = int(sys.argv[1]) if len(sys.argv) > 1 else 58
pr = pb.ricc_problem(N, 0.05)
ctx = _lib.Context(0)
ctx.set_operator((-pr.A - pr.Nc).T.tocsr(), pr.M.T.tocsr(), pr.J)
n, m = ctx.n, 16
nnz = pr.A.nnz  # placeholder, the library reports the unified count below
import scipy.sparse as sps
S = sps.bmat([[abs(pr.A) + abs(pr.M) + abs(pr.Nc), pr.J.T], [pr.J, None]], format="csr")
nnz = S.nnz
unit = 12 * nnz + 4 * (n + 1) + 16 * n * m
dev = torch.device("cuda", 0)
ms = pb.logshifts(1.0, 3e3, 16)
for G in (1, 2, 4, 8, 16):
    X = torch.randn(G, n, m, dtype=torch.float64, device=dev)
    Y = torch.empty_like(X)
    torch.cuda.synchronize()
    t = ctx.time_spmm_batch_dev(ms[:G], [1.0] * G, X.data_ptr(), m, Y.data_ptr(), 200)
    # correctness of the shared-value path against single launches
    err = 0.0
    for g in range(G):
        Y1 = torch.empty(n, m, dtype=torch.float64, device=dev)
        ctx.spmm_dev(float(ms[g]), 1.0, X[g].data_ptr(), m, Y1.data_ptr())
        ctx.synchronize()
        err = max(err, float((Y[g] - Y1).abs().max() / Y1.abs().max()))
    print("G=%2d: %.1f us/launch, %.2f us/unit, %.0f GB/s algorithmic (%.1f %% of 8 TB/s), max rel diff vs single %.1e"
          % (G, 1e3 * t, 1e3 * t / G, G * unit / (t * 1e-3) / 1e9, 100 * G * unit / (t * 1e-3) / 8e12, err), flush=True)
