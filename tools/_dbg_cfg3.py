import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optconpy_amd import backend, problems as pb
import sadptprj_riclyap_adi.lin_alg_utils as lau
import sadptprj_riclyap_adi.proj_ric_utils as pru
g = np.load("tests/golden/cfg3_golden.npz")
if os.environ.get("TOL"):
    backend.configure(gmres_tol=float(os.environ["TOL"]))
N, nu, alphau, NU, NY, ns, pmin, pmax = g["cfg"]
pr = pb.ricc_problem(int(N), float(nu), NU=int(NU), NY=int(NY), alphau=float(alphau))
mct = lau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=pr.mc_mat.T, transposedprj=True)
tb = lau.apply_invsqrt_fromright(pr.rmat, pr.b_mat, output="dense")
trct = lau.apply_invsqrt_fromright(pr.y_masmat, mct, output="dense")
ms = pb.logshifts(float(pmin), float(pmax), int(ns))
d = dict(pb.default_nwtn_adi_dict(), ms=ms, verbose=True, nwtn_max_steps=int(os.environ.get("NSTEPS", "7")))
if os.environ.get("SW"): d["sweep_width"] = int(os.environ["SW"])
F = (-pr.A - pr.Nc).tocsr()
out = pru.proj_alg_ric_newtonadi(mmat=pr.M, amat=F, jmat=pr.J, bmat=tb, wmat=trct, nwtn_adi_dict=d)
K = -pru.get_mTzzTtb(pr.M.T, out["zfac"], tb)
print({k: v for k, v in out.items() if k != "zfac"})
print("K err", np.linalg.norm(K - g["K_ric"]) / np.linalg.norm(g["K_ric"]), "oracle hist", g["upd_hist"].tolist())
