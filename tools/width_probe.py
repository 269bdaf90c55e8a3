"""Panel-width probe: time per GMRES iteration of one shift-solve as the panel widens
(is the cfg2 iteration latency-bound?).  python tools/width_probe.py [N] [m ...]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from optconpy_amd import _lib, problems as pb
N = int(sys.argv[1]) if len(sys.argv) > 1 else 58
ms = [int(a) for a in sys.argv[2:]] or [16, 32, 64, 128]
pr = pb.ricc_problem(N, 0.05)
ctx = _lib.Context(0)
ctx.set_operator((-pr.A - pr.Nc).T.tocsr(), pr.M.T.tocsr(), pr.J)
rng = np.random.default_rng(0)
for m in ms:
    R = rng.standard_normal((pr.NV, m))
    for p in (-30.0,):
        ctx.shift_solve(p, 1.0, R, strict=False)
        t0 = time.time()
        X, its, rr = ctx.shift_solve(p, 1.0, R, strict=False)
        dt = time.time() - t0
        print("m=%3d p=%g: its %d  %.1f ms  %.0f us/it  %.1f us/it/16cols" %
              (m, p, its, 1e3 * dt, 1e6 * dt / its, 1e6 * dt / its / (m / 16)), flush=True)
