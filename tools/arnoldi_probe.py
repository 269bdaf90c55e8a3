"""Arnoldi launch classes (dots / update+dots / update) of the batched GMRES at several basis sizes;
python tools/arnoldi_probe.py [N] [G] [reps]"""
import sys
sys.path.insert(0, ".")
import torch
from optconpy_amd import _lib, problems as pb
N = int(sys.argv[1]) if len(sys.argv) > 1 else 58
G = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
torch.cuda.set_device(0)
pr = pb.ricc_problem(N, 0.05)
ctx = _lib.Context(0)
ctx.set_operator((-pr.A - pr.Nc).T.tocsr(), pr.M.T.tocsr(), pr.J)
ms = pb.logshifts(1.0, 3e3, 16)[:G]
n = ctx.n
for nvec in (3, 7, 15, 29):
    line = "N=%d nvec=%2d:" % (N, nvec)
    for k, byts in (("dots", 128 + 32 * nvec), ("update_dots", 256 + 64 * nvec), ("update", 146 + 32 * nvec)):
        t = ctx.time_kernel_dev(k, ms, [1.0] * G, 16, nvec=nvec, reps=reps)
        line += "  %s %8.1f us (%.2f TB/s)" % (k, 1e3 * t, byts * n * G / (1e-3 * t) / 1e12)
    print(line, flush=True)
ctx.close()
