"""Developer probe: batched 16-shift solve iteration counts at one mesh size over operator variants
(NSE operator at several viscosities, the DRE operator) -- python tools/ml_probe.py N"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from optconpy_amd import _lib, problems as pb
N = int(sys.argv[1]) if len(sys.argv) > 1 else 106
dev = torch.device("cuda", 0)
m = 16
for nu, dre in ((0.05, False), (0.01, False), (0.0025, False), (0.0025, True)):
    pr = pb.ricc_problem(N, nu)
    MT = pr.M.T.tocsr()
    if dre:
        tau = float(np.diff(pb.get_tint(0.0, 1.0, 16, True)).max())
        calA = (-(0.5 * MT + tau * (pr.A.T + pr.Nc.T))).tocsr()
    else:
        calA = (-pr.A - pr.Nc).T.tocsr()
    ctx = _lib.Context(0)
    ctx.set_operator(calA, MT, pr.J)
    info = ctx.setup_info()
    W = torch.as_tensor(np.random.default_rng(0).standard_normal((pr.NV, m))).to(dev)
    ps = [float(p) for p in pb.logshifts(0.5 if dre else 1.0, 2e3, 16)]
    X = torch.empty(16, ctx.n, m, dtype=torch.float64, device=dev)
    for rep in range(2):
        ctx.clear_cache()
        t0 = time.time()
        its, rr = ctx.shift_solve_batch_dev(ps, [1.0] * 16, W.data_ptr(), 0, m, X.data_ptr(), strict=False)
        ctx.synchronize()
        dt = time.time() - t0
    print("N=%d nu=%g %s: levels %s dense %s | %.1f ms incl. setup, its %s, max relres %.1e"
          % (N, nu, "DRE" if dre else "NSE", info["levels"], info["dense_coarse"],
             1e3 * dt, its, rr.max()), flush=True)
    ctx.close()
