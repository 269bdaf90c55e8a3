"""Developer bring-up: exercise every layer of the HIP path against the oracle.

Run on a GPU box:  python tests/dev_bringup.py [N] [nu]   (developer script, not collected by pytest)
"""
import sys
import time

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, ".")
from optconpy_amd import _lib, problems as pb  # noqa: E402
from oracle import lin_alg_utils as olau, proj_ric_utils as opru  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 15
nu = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
stage = sys.argv[3] if len(sys.argv) > 3 else "all"


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


t0 = time.time()
pr = pb.ricc_problem(N, nu, alphau=1e-2)
M, A, J, Nc = pr.M, pr.A, pr.J, pr.Nc
NV, NP = pr.NV, pr.NP
n = NV + NP
print("problem N=%d NV=%d NP=%d  (%.1fs)" % (N, NV, NP, time.time() - t0), flush=True)
F = (-A - Nc).tocsr()
calA = F.T.tocsr()
calE = M.T.tocsr()

ctx = _lib.Context(0, verbose=1)
t0 = time.time()
ctx.set_operator(calA, calE, J)
print("set_operator %.2fs" % (time.time() - t0), flush=True)
rng = np.random.default_rng(0)

# 1. SpMM parity
for m in (16, 5, 33):
    X = rng.standard_normal((n, m))
    p = -3.0
    S = sps.bmat([[calA + p * calE, J.T], [J, None]], format="csr")
    Y = ctx.spmm(p, 1.0, X)
    print("spmm m=%d rel err %.2e" % (m, rel(Y, S @ X)), flush=True)

# 2. shift solves
for p in (-1.0, -10.0, -100.0, -1000.0):
    R = rng.standard_normal((NV, 16))
    t0 = time.time()
    X, its, rr = ctx.shift_solve(p, 1.0, R, strict=False)
    t1 = time.time() - t0
    lu = olau.SaddleLU(calA + p * calE, J)
    Xo = lu.solve(R)
    print("shift %8.1f: its %4d relres %.1e  err_v %.2e  (%.3fs)" %
          (p, its, rr.max(), rel(X[:NV], Xo[:NV]), t1), flush=True)

# projection solve
R = rng.standard_normal((NV, 8))
X, its, rr = ctx.shift_solve(1.0, 0.0, R, strict=False)
Xo = olau.SaddleLU(calE, J).solve(R)
print("proj solve: its %d relres %.1e err %.2e" % (its, rr.max(), rel(X[:NV], Xo[:NV])), flush=True)
if stage == "solve":
    sys.exit(0)

# 3. Lyapunov ADI
mct = olau.app_prj_via_sadpnt(amat=M, jmat=J, rhsv=pr.mc_mat.T, transposedprj=True)
tb = olau.apply_invsqrt_fromright(pr.rmat, pr.b_mat, output="sparse")
trct = olau.apply_invsqrt_fromright(pr.y_masmat, mct, output="dense")
ms = pb.logshifts(1.0, 3e3, 16)
d = dict(adi_max_steps=200, adi_newZ_reltol=1e-8, nwtn_max_steps=16, nwtn_upd_reltol=5e-8,
         nwtn_upd_abstol=1e-7, ms=ms, verbose=False)
prm = _lib.adi_params(d)
prm.verbose = 0
t0 = time.time()
Z, info = ctx.lyap_adi(ms, trct, prm)
tg = time.time() - t0
t0 = time.time()
Zo = opru.solve_proj_lyap_stein(amat=F, mmat=M, jmat=J, wmat=trct, adi_dict=d)
to = time.time() - t0
Kg = -(M.T @ (Z @ (Z.T @ tb.toarray())))
Ko = -opru.get_mTzzTtb(M.T, Zo["zfac"], tb)
print("lyap: gpu %s cols %d (%.2fs) | oracle steps %d (%.2fs) | K rel diff %.2e" %
      (info, Z.shape[1], tg, Zo["adi_steps"], to, rel(Kg, Ko)), flush=True)
Kg2 = -ctx.gain(tb.toarray(), Z=None)
print("gain kernel vs numpy: %.2e" % rel(Kg2, Kg), flush=True)

# compress
Zc, sv = ctx.compress(Z, thresh=1e-8, k=None)
Zco = opru.compress_Zsvd(Z, thresh=1e-8)
print("compress: k %d vs %d ; ||ZcZc^T - ZZ^T|| rel %.2e ; vs oracle %.2e" %
      (Zc.shape[1], Zco.shape[1],
       opru.comp_diff_zzt_fnorm(Zc, Z) / np.linalg.norm(Z.T @ Z),
       opru.comp_diff_zzt_fnorm(Zc, Zco) / np.linalg.norm(Z.T @ Z)), flush=True)
r2 = ctx.lyap_res_norm(Zc, trct)
r2o = opru.comp_proj_lyap_res_norm(Zc, F, M, trct, J)
print("res norm^2 gpu %.6e oracle %.6e" % (r2, r2o), flush=True)
if stage == "lyap":
    sys.exit(0)

# 4. Newton
t0 = time.time()
Zn, ninfo = ctx.ric_newtonadi(ms, tb.toarray(), trct, prm)
tg = time.time() - t0
t0 = time.time()
st = {}
on = opru.proj_alg_ric_newtonadi(mmat=M, amat=F, jmat=J, bmat=tb, wmat=trct, nwtn_adi_dict=d,
                                 stats=st)
to = time.time() - t0
Kg = -ctx.gain(tb.toarray(), Z=None)
Ko = -opru.get_mTzzTtb(M.T, on["zfac"], tb)
print("newton: gpu %s (%.2fs) | oracle steps %d %s (%.2fs) %s" %
      (ninfo, tg, on["nwtn_steps"], [round(u[1], 11) for u in on["upd_hist"]], to, st), flush=True)
print("K rel Frobenius diff %.3e   |K| %.4e" % (rel(Kg, Ko), np.linalg.norm(Ko)), flush=True)
