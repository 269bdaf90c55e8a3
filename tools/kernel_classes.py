"""One launch class of the batched GMRES per call of ricadi_time_kernel_dev, 16 groups, cfg2 (or
N given) -- the workload behind bench.py's `roofline_kernels`, for rocprofv3 --kernel-trace --stats.
python tools/kernel_classes.py [N] [G] [reps]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from optconpy_amd import _lib, problems as pb
N = int(sys.argv[1]) if len(sys.argv) > 1 else 58
G = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
torch.cuda.set_device(0)
pr = pb.ricc_problem(N, 0.05)
ctx = _lib.Context(0)
ctx.set_operator((-pr.A - pr.Nc).T.tocsr(), pr.M.T.tocsr(), pr.J)
ms = pb.logshifts(1.0, 3e3, 16)[:G]
for k in ("spmm", "dots", "update_dots", "update", "pc_restrict", "pc_coarse", "pc_sy_prows", "pc_two_term", "pc_jprod",
          "pc_schur", "pc_rect", "precond"):
    t = ctx.time_kernel_dev(k, ms, [1.0] * G, 16, nvec=7, reps=reps)
    print("%-12s %8.2f us per launch (HIP events, %d launches)" % (k, 1e3 * t, reps))
ctx.close()
