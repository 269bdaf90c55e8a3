"""Probe of the feed-forward saddle solve of the DRE sweep (solve_dae_ric.py:192-194) at cfg4 size:
operator M^T + tau (A+N)^T passed (i) as cal A with an empty cal E (what lau.solve_sadpnt_smw did in
round 2), (ii) split as cal E = M^T, cal A = tau (A+N)^T with (alpha, beta) = (1, 1).  Usage: ff_probe.py N tau"""
import os, sys, time
import numpy as np, scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optconpy_amd import _lib, problems as pb
N = int(sys.argv[1]) if len(sys.argv) > 1 else 106
tau = float(sys.argv[2]) if len(sys.argv) > 2 else 0.038
pr = pb.ricc_problem(N, 0.15 / 60.0)
MT = pr.M.T.tocsr()
AN = (pr.A + pr.Nc).T.tocsr()
at = (MT + tau * AN).tocsr()
rhs = np.random.default_rng(0).standard_normal((pr.NV, 1))
for tag, calA, calE, al, be in (("empty E", at, sps.csr_matrix(at.shape), 0.0, 1.0), ("split", (tau * AN).tocsr(), MT, 1.0, 1.0)):
    for m in (1, 16):
        with _lib.Context(0, verbose=0, gmres_maxit=400) as ctx:
            ctx.set_operator(calA, calE, pr.J)
            R = np.random.default_rng(1).standard_normal((pr.NV, m))
            t0 = time.time()
            X, its, rr = ctx.shift_solve(al, be, R, strict=False)
            print("%-8s m=%2d levels %s: iters %d worst relres %.2e (%.2f s)" % (tag, m, ctx.setup_info()["levels"], its, rr.max(), time.time() - t0), flush=True)
