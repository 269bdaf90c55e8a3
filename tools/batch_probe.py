"""Batched vs per-shift solves at cfg2: wall time of one sweep of G shifts.  python tools/batch_probe.py [N]"""
import os, sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from optconpy_amd import _lib, problems as pb
N = int(sys.argv[1]) if len(sys.argv) > 1 else 58
pr = pb.ricc_problem(N, 0.05)
ctx = _lib.Context(0)
ctx.set_operator((-pr.A - pr.Nc).T.tocsr(), pr.M.T.tocsr(), pr.J)
rng = np.random.default_rng(0)
m = 16
dev = torch.device("cuda", 0)
W = torch.as_tensor(rng.standard_normal((pr.NV, m))).to(dev)
ms = pb.logshifts(1.0, 1e3, 16)
for G in ([int(a) for a in sys.argv[2:]] or [1, 2, 4, 8, 16]):
    ps = [float(p) for p in ms[:G]]
    X = torch.empty(G, ctx.n, m, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.time()
        its, rr = ctx.shift_solve_batch_dev(ps, [1.0] * G, W.data_ptr(), 0, m, X.data_ptr(), strict=False)
        ctx.synchronize()
        dt = time.time() - t0
    if os.environ.get("BATCH_ONLY"):
        print("G=%2d: batched %.1f ms (its %s, max relres %.1e) | %.1f us per group-iteration, %.0f us per lockstep iteration"
              % (G, 1e3 * dt, its, rr.max(), 1e6 * dt / max(sum(its), 1), 1e6 * dt / max(its)), flush=True)
        continue
    t0 = time.time()
    it1 = 0
    Y = torch.empty(ctx.n, m, dtype=torch.float64, device=dev)
    for p in ps:
        i, _ = ctx.shift_solve_dev(p, 1.0, W.data_ptr(), m, Y.data_ptr(), strict=False)
        it1 += i
    ctx.synchronize()
    dt1 = time.time() - t0
    print("G=%2d: batched %.1f ms (its %s, max relres %.1e) | one by one %.1f ms (its %d) | %.1f us per group-iteration vs %.1f"
          % (G, 1e3 * dt, its, rr.max(), 1e3 * dt1, it1, 1e6 * dt / max(sum(its), 1), 1e6 * dt1 / max(it1, 1)), flush=True)
