"""Time the block QR (K5) on a random NV x c matrix:  python tools/qr_probe.py 26450 456
(RICADI_QR_PANEL=32: round-2 panels of 32 columns)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optconpy_amd import _lib  # noqa: E402

nv, c = int(sys.argv[1]), int(sys.argv[2])
ctx = _lib.Context(0)
ctx.set_dims(nv)
z = torch.randn(nv, c, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
ctx.time_qr_dev(z.data_ptr(), c, 2)
ms = min(ctx.time_qr_dev(z.data_ptr(), c, 10) for _ in range(3))
fl = 2.0 * nv * c * c - 2.0 / 3.0 * c ** 3
print("block QR %d x %d, panel %s: %.3f ms, %.2f TFLOP/s algorithmic = %.3f of 78.6" % (
    nv, c, os.environ.get("RICADI_QR_PANEL", "128"), ms, fl / ms / 1e9, fl / ms / 1e9 / 78.6))
Z = z.cpu().numpy()
Q, R = ctx.qr(Z)
print("  ||QR - Z|| / ||Z|| %.2e   ||Q^T Q - I|| %.2e   lower(R) %.1e" % (
    np.linalg.norm(Q @ R - Z) / np.linalg.norm(Z), np.linalg.norm(Q.T @ Q - np.eye(c)), np.abs(np.tril(R, -1)).max()))
ctx.close()
