"""Only batched tile-SpMM launches (for rocprofv3 --pmc passes).
python tools/spmm_batch_pmc.py N G [reps]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from optconpy_amd import _lib, problems as pb
N, G = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
pr = pb.ricc_problem(N, {75: 0.15 / 40, 106: 0.15 / 60}.get(N, 0.05))
ctx = _lib.Context(0, use_coarse=0)         # no coarse inverse needed for the SpMM itself
ctx.set_operator((-pr.A - pr.Nc).T.tocsr(), pr.M.T.tocsr(), pr.J)
n, m = ctx.n, 16
ms = pb.logshifts(1.0, 3e3, 16)
X = torch.randn(G, n, m, dtype=torch.float64, device="cuda")
Y = torch.empty_like(X)
torch.cuda.synchronize()
t = ctx.time_spmm_batch_dev(ms[:G], [1.0] * G, X.data_ptr(), m, Y.data_ptr(), reps)
print("N=%d n=%d G=%d: %.1f us per launch" % (N, n, G, 1e3 * t))
