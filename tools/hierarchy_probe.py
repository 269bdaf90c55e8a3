"""One batched 16-shift solve at mesh N with a given coarse_max: hierarchy, iterations, worst residual."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from optconpy_amd import _lib, problems as pb
N, cm = int(sys.argv[1]), int(sys.argv[2])
nu = float(os.environ.get('NU', '0.05'))
pr = pb.ricc_problem(N, nu)
kw = {}
if len(sys.argv) > 4: kw = dict(agg_v=int(sys.argv[3]), agg_p=int(sys.argv[4]))
ctx = _lib.Context(0, coarse_max=cm, gmres_maxit=600, verbose=1, **kw)
ctx.set_operator((-pr.A - pr.Nc).T.tocsr(), pr.M.T.tocsr(), pr.J)
info = ctx.setup_info()
m = 16
W = torch.as_tensor(np.random.default_rng(0).standard_normal((pr.NV, m))).cuda()
ps = [float(p) for p in pb.logshifts(1.0, 3e3, 16)]
X = torch.empty(16, ctx.n, m, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
t0 = time.time()
its, rr = ctx.shift_solve_batch_dev(ps, [1.0] * 16, W.data_ptr(), 0, m, X.data_ptr(), strict=False)
ctx.synchronize()
print("nu=%g" % nu, "N=%d coarse_max=%d: levels %d kc %d dense %d | its %s | worst relres %.2e | %.1f ms"
      % (N, cm, info["levels"], info["kc"], info["dense_coarse"], its, rr.max(), 1e3 * (time.time() - t0)), flush=True)
