"""A/B timing of the SpMM variants (separate processes: the variant is read once)."""
import os, sys, json
sys.path.insert(0, ".")
import numpy as np, torch
from optconpy_amd import _lib, problems as pb
N = int(sys.argv[1]); m = int(sys.argv[2])
pr = pb.ricc_problem(N, 0.05)
ctx = _lib.Context(0, use_coarse=0, verbose=1)
calA = (-pr.A - pr.Nc).T.tocsr()
ctx.set_operator(calA, pr.M.T.tocsr(), pr.J)
n = pr.NV + pr.NP
nnz = (calA + pr.M).nnz + 2 * pr.J.nnz
x = torch.randn(n, m, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
torch.cuda.synchronize()
ctx.time_spmm_dev(-3.0, 1.0, x.data_ptr(), m, y.data_ptr(), 20)
best = min(ctx.time_spmm_dev(-3.0, 1.0, x.data_ptr(), m, y.data_ptr(), 200) for _ in range(5))
nbytes = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n * m
import scipy.sparse as sps
S = sps.bmat([[calA - 3.0 * pr.M, pr.J.T], [pr.J, None]], format="csr")
err = np.linalg.norm(y.cpu().numpy() - S @ x.cpu().numpy()) / np.linalg.norm(S @ x.cpu().numpy())
print("variant %s N=%d m=%d: %.2f us  %.0f GB/s (%.1f%% of 8 TB/s)  err %.1e" %
      (os.environ.get("RICADI_SPMM", "2"), N, m, best * 1e3, nbytes / best / 1e6, nbytes / best / 1e6 / 80, err))
