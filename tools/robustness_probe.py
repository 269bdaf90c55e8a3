"""Shift-solve iteration counts and K parity over problem variants (needs the oracle: dev tool)."""
import sys, time, warnings
sys.path.insert(0, ".")
import numpy as np
from optconpy_amd import _lib, problems as pb
from oracle import lin_alg_utils as olau, proj_ric_utils as opru

def rel(a, b): return np.linalg.norm(a - b) / np.linalg.norm(b)

for N, nu, order in ((15, 0.1, "interleaved"), (15, 0.02, "component"), (15, 0.005, "component"),
                     (30, 0.02, "component"), (30, 0.005, "interleaved")):
    pr = pb.ricc_problem(N, nu, ordering=order)
    F = (-pr.A - pr.Nc).tocsr()
    ctx = _lib.Context(0)
    ctx.set_operator(F.T.tocsr(), pr.M.T.tocsr(), pr.J)
    rng = np.random.default_rng(0)
    R = rng.standard_normal((pr.NV, 16))
    line = "N=%d nu=%g %s:" % (N, nu, order)
    for p in (-1.0, -30.0, -1000.0):
        with warnings.catch_warnings(record=True):
            X, its, rr = ctx.shift_solve(p, 1.0, R, strict=False)
        line += "  p=%g its %d (%.0e)" % (p, its, rr.max())
    # DRE-like operator: -(M/2 + tau (A+N)^T), tau = 0.01, shifts ~ -1
    tau = 0.01
    ft = (-(0.5 * pr.M.T + tau * (pr.A + pr.Nc).T)).tocsr()
    c2 = _lib.Context(0)
    c2.set_operator(ft, pr.M.T.tocsr(), pr.J)
    X, its, rr = c2.shift_solve(-1.0, 1.0, R, strict=False)
    line += "  | DRE tau=.01 p=-1 its %d (%.0e)" % (its, rr.max())
    c2.close()
    if N == 15:
        mct = olau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=pr.mc_mat.T, transposedprj=True)
        tb = olau.apply_invsqrt_fromright(pr.rmat, pr.b_mat, output="dense")
        trct = olau.apply_invsqrt_fromright(pr.y_masmat, mct, output="dense")
        ms = pb.logshifts(1.0, 1e3, 8)
        d = dict(pb.default_nwtn_adi_dict(), ms=ms)
        Z, info = ctx.ric_newtonadi(ms, tb, trct, _lib.adi_params(d))
        K = -ctx.gain(tb)
        ref = opru.proj_alg_ric_newtonadi(mmat=pr.M, amat=F, jmat=pr.J, bmat=tb, wmat=trct, nwtn_adi_dict=d)
        Ko = -opru.get_mTzzTtb(pr.M.T, ref["zfac"], tb)
        line += "  | Newton steps %d/%d K rel diff %.1e" % (info["nwtn_steps"], ref["nwtn_steps"], rel(K, Ko))
    print(line, flush=True)
    ctx.close()
