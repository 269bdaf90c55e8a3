"""Hard / easy operators under the current storage switches (set RICADI_BASIS16 / RICADI_BASIS64 outside).
python tools/hard_probe.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from optconpy_amd import _lib, problems as pb
rng = np.random.default_rng(0)
for N, nu in ((15, 0.005), (30, 0.005), (58, 0.01), (58, 0.05)):
    pr = pb.ricc_problem(N, nu)
    R = rng.standard_normal((pr.NV, 16))
    ctx = _lib.Context(0, gmres_maxit=3000)
    ctx.set_operator((-pr.A - pr.Nc).T.tocsr(), pr.M.T.tocsr(), pr.J)
    out = []
    for p in (-1.0, -30.0, -1000.0):
        ctx.shift_solve(p, 1.0, R, strict=False)
        t0 = time.time()
        X, its, rr = ctx.shift_solve(p, 1.0, R, strict=False)
        out.append("p=%g: %d its %.1f ms (%.0e)" % (p, its, 1e3 * (time.time() - t0), rr.max()))
    print("N=%d nu=%g | %s" % (N, nu, " | ".join(out)), flush=True)
    ctx.close()
