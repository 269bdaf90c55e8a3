"""Time the internal recompression (ricadi_recompress) on a graded NV x c factor; run once per route:
    python tools/recompress_probe.py 26450 768            (pivoted Cholesky, the default)
    RICADI_RECOMPRESS_EIG=1 python tools/recompress_probe.py 26450 768   (Gram + rocSOLVER dsyevd)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optconpy_amd import _lib  # noqa: E402

nv, c = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1)
Q = rng.standard_normal((nv, c))
Q /= np.linalg.norm(Q, axis=0)
V, _ = np.linalg.qr(rng.standard_normal((c, c)))
Z = (Q * np.logspace(0, -14, c)) @ V.T
ctx = _lib.Context(0)
ctx.set_dims(nv)
Zc = ctx.recompress(Z)
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    Zc = ctx.recompress(Z)
dt = (time.perf_counter() - t0) / reps
G = Z.T @ Z
X = Z.T @ Zc
d2 = np.linalg.norm(Zc @ Zc.T - Z @ Z.T) ** 2 if nv <= 4000 else float("nan")
s = np.linalg.svd(Z, compute_uv=False)
print("route %s  nv %d c %d -> k %d (optimal at 3e-8: %d)  %.2f ms per call incl. PCIe  err^2/||G||^2 %.2e" % (
    "eig" if os.environ.get("RICADI_RECOMPRESS_EIG") else "pchol", nv, c, Zc.shape[1],
    int((s > 3e-8 * s[0]).sum()), dt * 1e3, d2 / np.linalg.norm(G) ** 2))
ctx.close()
