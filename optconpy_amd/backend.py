"""Process-wide HIP context management for the host-side mirror.

The reference is single-threaded and re-passes its matrices with every call
(``solve_dae_ric.py:152-159,192-194``); to keep that calling convention cheap
one :class:`optconpy_amd._lib.Context` per process is kept alive and the
operator is re-uploaded only when its fingerprint changes.
"""
from __future__ import annotations

import os

import numpy as np
import scipy.sparse as sps

from . import _lib

_ctx = None
_ctx_key = None
_opts = {}


def configure(**opts):
    """Set inner-solver options (see ``ricadi_opts`` in include/ricadi.h).

    Takes effect for the next operator; drops the cached context.
    """
    global _opts
    _opts = dict(opts)
    reset()


def reset():
    global _ctx, _ctx_key
    if _ctx is not None:
        _ctx.close()
    _ctx = None
    _ctx_key = None


def device_id():
    return int(os.environ.get("LOCAL_RANK", os.environ.get("RICADI_DEVICE", "0")))


try:                                            # 128-bit non-cryptographic hash, ~10 GB/s
    import xxhash

    def _hasher():
        return xxhash.xxh3_128()
except ImportError:                             # pragma: no cover - the image ships xxhash
    import hashlib

    def _hasher():
        return hashlib.blake2b(digest_size=16)


def content_hash(m):
    """EXACT content key of a scipy sparse matrix as handed over: format, shape and a 128-bit hash of the bytes of
    its index and value arrays.  The reference re-passes its matrices with every call; whether the resident
    operator can be reused is decided on this key -- two different matrices never share it (up to a 2^-128
    collision), whatever their sums.  ~1 ms per 10 MB."""
    h = _hasher()
    for name in ("indptr", "indices", "data", "row", "col", "offsets"):
        a = getattr(m, name, None)
        if a is not None:
            a = np.ascontiguousarray(a)
            h.update(name.encode())
            h.update(str(a.dtype).encode())
            h.update(a.view(np.uint8).data if a.size else b"")
    return (getattr(m, "format", type(m).__name__), tuple(m.shape), int(m.nnz), h.hexdigest())


def _fingerprint(m):
    if m is None:
        return None
    fp = getattr(m, "_ricadi_fp", None)        # the mirror's own converted operands carry theirs (proj_ric_utils._orient)
    if fp is not None:
        return fp
    if not sps.issparse(m):
        m = sps.csr_matrix(m)
    if not (sps.isspmatrix_csr(m) and m.has_canonical_format):
        m = sps.csr_matrix(m, copy=True)        # the key is that of the canonical CSR form the library receives
        m.sum_duplicates()
    return content_hash(m)


def context():
    """The process-wide context (created on first use; needs a GPU)."""
    global _ctx
    if _ctx is None:
        _ctx = _lib.Context(device_id(), **_opts)
    return _ctx


_mass_hint = None


def mass_hint(nv):
    """The last mass-like ``cal E`` (more than two entries per row) an operator was set with, if it has
    ``nv`` rows.  ``lau.solve_sadpnt_smw`` receives ONE matrix (``M^T + tau (A+N)^T``,
    ``solve_dae_ric.py:173,192-194``); with the mass matrix of the Riccati solve of the same time step as
    ``cal E`` -- at coefficient alpha = 0, so it does not enter the operator -- the preconditioner's
    aggregates follow the mass matrix's graph (velocity components kept apart) instead of the union
    pattern: N = 58: 92 instead of 215-285 GMRES iterations, and the three-level hierarchy of n ~ 1e5
    works at all (it stagnates on mixed-component aggregates)."""
    if _mass_hint is not None and _mass_hint.shape[0] == nv:
        return _mass_hint
    return None


def context_for(calA, calE, J):
    """Context with the operator ``[[beta*calA + alpha*calE, J^T],[J,0]]`` set."""
    global _ctx_key, _mass_hint
    if calE is not None and sps.issparse(calE) and calE.nnz > 2 * calE.shape[0]:
        _mass_hint = calE
    key = (_fingerprint(calA), _fingerprint(calE), _fingerprint(J))
    ctx = context()
    if key != _ctx_key:
        ctx.set_operator(calA, calE, J)
        _ctx_key = key
    return ctx


def operator_has_cale(MT):
    """Whether the resident operator's ``cal E`` is the matrix ``MT`` (by fingerprint): the device-level gain
    (``ricadi_gain_dev``) multiplies by the context's ``cal E``."""
    return (_ctx is not None and isinstance(_ctx_key, tuple) and len(_ctx_key) == 3 and _ctx_key[0] != "dims"
            and _ctx_key[1] == _fingerprint(MT))


def context_dims(nv):
    """Context that only knows NV (compression, explicit-matrix gain)."""
    global _ctx_key
    ctx = context()
    if ctx.nv != nv:
        ctx.set_dims(nv)
        _ctx_key = ("dims", nv)
    return ctx


def ensure_exchange(ctx, panel_cols, nshifts, group=None):
    """Under ``torch.distributed`` (one process per GPU, world size > 1) the ADI sweeps of the
    drop-in shard by shift over the ranks (``ricadi_set_exchange``): install / resize the
    all-gather buffers of ``ctx`` for panels of ``panel_cols`` columns.  ``RICADI_SHIFT_PARALLEL=0``
    keeps every rank on the whole shift list (replicas).  No-op in a single process."""
    try:
        import torch.distributed as dist
    except ImportError:
        return
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1 \
        and os.environ.get("RICADI_SHIFT_PARALLEL", "1") != "0"
    cur = getattr(ctx, "_xchg", None)
    if not on:
        if cur is not None:
            ctx.set_exchange(False)
        return
    world = dist.get_world_size(group)
    per_rank = min(16, -(-int(nshifts) // world) + 1)
    need = per_rank * ctx.n * int(panel_cols) + 512
    if cur is None or cur[3] is not group or cur[4] < need:
        ctx.set_exchange(group, panel_cols=int(panel_cols), per_rank=per_rank)
