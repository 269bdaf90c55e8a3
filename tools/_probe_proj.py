"""Iterations of the projection solve S(1, 0) = [[M^T, J^T],[J, 0]] with the hierarchy chosen for cal A."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optconpy_amd import _lib, problems as pb
N = int(sys.argv[1]); kw = {}
for a in sys.argv[2:]:
    k, v = a.split("="); kw[k] = int(v)
pr = pb.ricc_problem(N, float(os.environ.get("NU", "0.05")))
ctx = _lib.Context(0, **kw)
ctx.set_operator((-pr.A - pr.Nc).T.tocsr(), pr.M.T.tocsr(), pr.J)
info = ctx.setup_info()
R = np.random.default_rng(1234).standard_normal((pr.NV, 16))
t0 = time.time()
X, it, rr = ctx.shift_solve(1.0, 0.0, R, strict=False)
print("N=%d %s: levels %d kc %d | projection solve %d iterations, worst relres %.1e, %.0f ms" % (N, kw, info["levels"], info["kc"], it, rr.max(), 1e3 * (time.time() - t0)), flush=True)
