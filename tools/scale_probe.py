"""Shift-solve scaling probe on the larger BASELINE configurations (no oracle:
the library's true residual is the check).  python tools/scale_probe.py N [coarse_max]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from optconpy_amd import _lib, problems as pb
N = int(sys.argv[1])
cm = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
restart = int(sys.argv[3]) if len(sys.argv) > 3 else 60
nu = {75: 0.15 / 40, 106: 0.15 / 60}.get(N, 0.05)
t0 = time.time()
pr = pb.ricc_problem(N, nu)
print("N=%d nu=%g NV=%d NP=%d assembled in %.1fs" % (N, nu, pr.NV, pr.NP, time.time() - t0), flush=True)
ctx = _lib.Context(0, verbose=1, coarse_max=cm, gmres_maxit=1500, gmres_restart=restart)
t0 = time.time()
ctx.set_operator((-pr.A - pr.Nc).T.tocsr(), pr.M.T.tocsr(), pr.J)
print("set_operator %.1fs" % (time.time() - t0), flush=True)
rng = np.random.default_rng(0)
R = rng.standard_normal((pr.NV, 16))
for p in (-1000.0, -100.0, -10.0, -1.0):
    t0 = time.time()
    X, its, rr = ctx.shift_solve(p, 1.0, R, strict=False)
    t1 = time.time() - t0
    t0 = time.time()
    X, its2, rr2 = ctx.shift_solve(p, 1.0, R, strict=False)     # preconditioner cached now
    print("p=%8.1f: its %4d relres %.1e  first %.2fs  cached %.2fs (%.0f us/it)" %
          (p, its, rr.max(), t1, time.time() - t0, 1e6 * (time.time() - t0) / max(its2, 1)), flush=True)
