"""ctypes binding of ``libricadi_hip.so`` (the C-ABI in ``include/ricadi.h``).

The shared library is the product; there is no CPU fallback.  Importing this
module only loads the library; creating a :class:`Context` needs a visible
MI355X and raises ``RuntimeError`` otherwise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import scipy.sparse as sps

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RICADI_LIB", os.path.join(_HERE, "libricadi_hip.so"))

RICADI_OK = 0
RICADI_ENOCONV = -3
MAX_M = 128
# ricadi_version() this mirror was written for: the stats arrays' lengths and the meaning of their slots
# are part of the ABI and are not covered by the struct handshake below
ABI_VERSION = 401


class RicadiOpts(C.Structure):
    _fields_ = [("gmres_tol", C.c_double), ("gmres_restart", C.c_int),
                ("gmres_maxit", C.c_int), ("bj_block", C.c_int), ("agg_v", C.c_int),
                ("agg_p", C.c_int), ("coarse_max", C.c_int), ("use_coarse", C.c_int),
                ("max_levels", C.c_int), ("verbose", C.c_int), ("compress_qr", C.c_int)]


class RicadiAdiParams(C.Structure):
    _fields_ = [("adi_max_steps", C.c_int), ("adi_newZ_reltol", C.c_double),
                ("nwtn_max_steps", C.c_int), ("nwtn_upd_reltol", C.c_double),
                ("nwtn_upd_abstol", C.c_double), ("project_w", C.c_int),
                ("verbose", C.c_int), ("compress_cols", C.c_int), ("sweep_width", C.c_int)]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol declared in include/ricadi.h
SIGNATURES = {
    "ricadi_last_error": (C.c_char_p, []),
    "ricadi_version": (C.c_int, []),
    "ricadi_sizeof_opts": (C.c_int, []),
    "ricadi_sizeof_adi_params": (C.c_int, []),
    "ricadi_struct_signature": (C.c_char_p, []),
    "ricadi_default_opts": (None, [C.POINTER(RicadiOpts)]),
    "ricadi_default_adi_params": (None, [C.POINTER(RicadiAdiParams)]),
    "ricadi_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "ricadi_destroy": (C.c_int, [_vp]),
    "ricadi_set_opts": (C.c_int, [_vp, C.POINTER(RicadiOpts)]),
    "ricadi_stream": (_vp, [_vp]),
    "ricadi_synchronize": (C.c_int, [_vp]),
    "ricadi_set_operator": (C.c_int, [_vp, C.c_int, C.c_int, _ip, _ip, _dp, _ip, _ip, _dp,
                                      _ip, _ip, _dp]),
    "ricadi_clear_cache": (C.c_int, [_vp]),
    "ricadi_set_dims": (C.c_int, [_vp, C.c_int]),
    "ricadi_set_lowrank": (C.c_int, [_vp, _dp, _dp, C.c_int]),
    "ricadi_spmm": (C.c_int, [_vp, C.c_double, C.c_double, _dp, C.c_int, _dp]),
    "ricadi_precond_apply": (C.c_int, [_vp, C.c_double, C.c_double, _dp, C.c_int, _dp]),
    "ricadi_shift_solve": (C.c_int, [_vp, C.c_double, C.c_double, _dp, _dp, C.c_int, _dp,
                                     C.POINTER(C.c_int), _dp]),
    "ricadi_lyap_adi": (C.c_int, [_vp, _dp, C.c_int, _dp, C.c_int, C.POINTER(RicadiAdiParams),
                                  _dp, C.POINTER(C.c_int), _dp]),
    "ricadi_ric_newtonadi": (C.c_int, [_vp, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp,
                                       C.c_int, _dp, C.POINTER(RicadiAdiParams), _dp, C.c_int,
                                       C.POINTER(C.c_int), _dp]),
    "ricadi_compress": (C.c_int, [_vp, _dp, C.c_int, C.c_double, C.c_int, _dp,
                                  C.POINTER(C.c_int), _dp]),
    "ricadi_recompress": (C.c_int, [_vp, _dp, C.c_int, C.c_double, _dp, C.POINTER(C.c_int)]),
    "ricadi_gain": (C.c_int, [_vp, _ip, _ip, _dp, _dp, C.c_int, _dp, C.c_int, _dp]),
    "ricadi_lyap_res_norm": (C.c_int, [_vp, _dp, C.c_int, _dp, C.c_int, _dp]),
    "ricadi_factor_cols": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "ricadi_factor_get": (C.c_int, [_vp, _dp, C.c_int]),
    "ricadi_factor_set": (C.c_int, [_vp, _dp, C.c_int]),
    "ricadi_factor_get_dev": (C.c_int, [_vp, _vp, C.c_int]),
    "ricadi_ric_newtonadi_dev": (C.c_int, [_vp, _dp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp,
                                           C.POINTER(RicadiAdiParams), C.POINTER(C.c_int), _dp]),
    "ricadi_spmm_dev": (C.c_int, [_vp, C.c_double, C.c_double, _vp, C.c_int, _vp]),
    "ricadi_shift_solve_dev": (C.c_int, [_vp, C.c_double, C.c_double, _vp, C.c_int, _vp,
                                         C.POINTER(C.c_int), _dp]),
    "ricadi_shift_solve_batch_dev": (C.c_int, [_vp, C.c_int, _dp, _dp, _vp, C.c_int64, C.c_int, _vp,
                                               C.POINTER(C.c_int), _dp]),
    "ricadi_time_spmm_batch_dev": (C.c_int, [_vp, C.c_int, _dp, _dp, _vp, C.c_int, _vp, C.c_int,
                                             C.POINTER(C.c_double)]),
    "ricadi_sweep_recombine_dev": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _dp, _dp, _vp, _vp,
                                             C.POINTER(C.c_double)]),
    "ricadi_sweep_recombine_slots_dev": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int, _dp, _dp, _vp, _vp,
                                                   C.POINTER(C.c_double), _dp]),
    "ricadi_apply_e_dev": (C.c_int, [_vp, C.c_double, _vp, C.c_int, _vp]),
    "ricadi_lincomb_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int64, _dp, _vp]),
    "ricadi_gain_dev": (C.c_int, [_vp, C.c_double, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp]),
    "ricadi_panel_norms_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _dp, _dp]),
    "ricadi_time_spmm_dev": (C.c_int, [_vp, C.c_double, C.c_double, _vp, C.c_int, _vp, C.c_int,
                                       _dp]),
    "ricadi_time_kernel_dev": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(C.c_double)]),
    "ricadi_qr": (C.c_int, [_vp, _dp, C.c_int, _dp, _dp]),
    "ricadi_setup_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.c_int]),
    "ricadi_time_qr_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "ricadi_time_gram_dev": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _dp]),
    "ricadi_set_recycle": (C.c_int, [_vp, C.c_int]),
    "ricadi_set_exchange": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int64]),
    "ricadi_rccl_unique_id": (C.c_int, [_vp, C.c_int]),
    "ricadi_set_exchange_rccl": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, C.c_int64]),
    "ricadi_exchange_count": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "ricadi_dense_inverse_batch": (C.c_int, [_vp, C.c_int, C.c_int, _dp, C.POINTER(C.c_int)]),
    "ricadi_host_deal": (C.c_int, [_dp, C.c_int, C.c_int, _ip]),
    "ricadi_host_sa_criterion": (C.c_int, [C.c_int, _ip, _ip, _dp, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                           C.POINTER(C.c_int)]),
    "ricadi_host_plan_levels": (C.c_int, [C.c_int, C.c_int, _ip, _ip, _dp, _ip, _ip, _dp, _ip, _ip, _dp,
                                          C.POINTER(RicadiOpts), _ip]),
    "ricadi_host_aggregate": (C.c_int, [C.c_int, _ip, _ip, C.c_int, _ip]),
    "ricadi_host_cauchy": (C.c_int, [_dp, C.c_int, _dp, _dp]),
}

EXCHANGE_FN = C.CFUNCTYPE(C.c_int, _vp, _vp, _vp, C.c_int64)

_lib = None


def load():
    """Load the shared library (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libricadi_hip.so is missing ({0}); build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` -- there is no "
            "CPU fallback for the HIP path".format(LIB_PATH))
    # torch wheels bundle their own ROCm runtime (libamdhip64 / rocblas /
    # rocsolver, same SONAMEs as /opt/rocm).  Two HIP runtimes in one process
    # crash, so torch -- needed anyway for torch.distributed -- is imported
    # first and our NEEDED entries then bind to the already loaded copies.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.ricadi_version() != ABI_VERSION:
        raise RuntimeError("{0}: library ABI version {1}, optconpy_amd/_lib.py was written for {2} -- rebuild "
                           "the library (__graft_entry__.build()) or update the mirror"
                           .format(LIB_PATH, lib.ricadi_version(), ABI_VERSION))
    # struct handshake: the library reads every field of the structs it is handed, so a
    # mirror that is shorter than the library's struct makes it read past our buffer
    # (root cause of the round-1 abort: a rebuilt .so with a new ricadi_adi_params field
    # met a not-yet-updated mirror).  Refuse to run with a mismatched build.
    for what, ours, theirs in (("ricadi_opts", C.sizeof(RicadiOpts), lib.ricadi_sizeof_opts()),
                               ("ricadi_adi_params", C.sizeof(RicadiAdiParams),
                                lib.ricadi_sizeof_adi_params())):
        if ours != theirs:
            raise RuntimeError("{0}: struct {1} is {2} bytes in the library but {3} bytes in "
                               "optconpy_amd/_lib.py -- rebuild the library (__graft_entry__.build())"
                               .format(LIB_PATH, what, theirs, ours))
    ours = ";".join("{0}:{1}".format(nm, "".join("d" if t is C.c_double else "i" for _, t in st._fields_))
                    for nm, st in (("ricadi_opts", RicadiOpts), ("ricadi_adi_params", RicadiAdiParams)))
    theirs = lib.ricadi_struct_signature().decode()
    if ours != theirs:
        raise RuntimeError("{0}: struct layout '{1}' in the library, '{2}' in optconpy_amd/_lib.py "
                           "-- rebuild the library or update the mirror".format(LIB_PATH, theirs, ours))
    _lib = lib
    return lib


def _chk(rc, allow_noconv=False):
    if rc == RICADI_OK or (allow_noconv and rc == RICADI_ENOCONV):
        return rc
    msg = load().ricadi_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError("ricadi: " + msg)
    raise RuntimeError("ricadi error {0}: {1}".format(rc, msg))


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def as_panel(a, nrows=None):
    """C-contiguous float64 2-D array (column vectors become n x 1)."""
    if sps.issparse(a):
        a = a.toarray()
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    a = np.ascontiguousarray(a)
    if nrows is not None and a.shape[0] != nrows:
        raise ValueError("panel has {0} rows, expected {1}".format(a.shape[0], nrows))
    return a


def as_csr(a):
    """Sorted CSR with int32 indices and float64 values (input csr / csc / dense)."""
    m = sps.csr_matrix(a, dtype=np.float64)
    m.sum_duplicates()
    m.sort_indices()
    if m.nnz >= 2 ** 31:
        raise ValueError("matrix too large for int32 indices")
    return (np.ascontiguousarray(m.indptr, dtype=np.int32),
            np.ascontiguousarray(m.indices, dtype=np.int32),
            np.ascontiguousarray(m.data, dtype=np.float64), m.shape)


def _warn_nonconverged(info):
    """Non-convergence of an inner solve is not an error at the C-ABI (the outer
    loops go on, like the reference's when it hits *_max_steps) -- but say so."""
    if info.get("gmres_nonconverged", 0):
        import warnings
        warnings.warn("ricadi: {0} of {1} shift-solves stopped at gmres_maxit above the tolerance "
                      "(worst relative residual {2:.1e}); the low-rank factor may be inaccurate"
                      .format(info["gmres_nonconverged"], info["shift_solves"],
                              info["gmres_worst_relres"]), RuntimeWarning, stacklevel=3)


def default_opts(**kw):
    o = RicadiOpts()
    load().ricadi_default_opts(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise KeyError("unknown ricadi option " + k)
        setattr(o, k, v)
    return o


def adi_params(d=None, project_w=True):
    """Translate the reference's ``nwtn_adi_dict`` (optcont_main.py:122-131).

    Unknown keys are ignored, missing ones take the reference defaults.
    """
    p = RicadiAdiParams()
    load().ricadi_default_adi_params(C.byref(p))
    d = {} if d is None else d
    for k in ("adi_max_steps", "nwtn_max_steps"):
        if k in d:
            setattr(p, k, int(d[k]))
    for k in ("adi_newZ_reltol", "nwtn_upd_reltol", "nwtn_upd_abstol"):
        if k in d:
            setattr(p, k, float(d[k]))
    p.verbose = 1 if d.get("verbose", False) else 0
    p.project_w = 1 if d.get("project_w", project_w) else 0
    p.compress_cols = int(d.get("compress_cols", 0))
    p.sweep_width = int(d.get("sweep_width", 1))
    return p


class Context:
    """One GPU, one stream, one saddle-point operator."""

    def __init__(self, device=0, **opts):
        self._lib = load()
        self._h = _vp()
        _chk(self._lib.ricadi_create(int(device), C.byref(self._h)))
        self.nv = self.np_ = self.n = 0
        if opts:
            self.set_opts(**opts)

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.ricadi_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def stream(self):
        return self._lib.ricadi_stream(self._h)

    def synchronize(self):
        _chk(self._lib.ricadi_synchronize(self._h))

    def set_opts(self, **kw):
        o = default_opts(**kw)
        self._opts = o
        _chk(self._lib.ricadi_set_opts(self._h, C.byref(o)))

    # -- operator ---------------------------------------------------------
    def set_operator(self, calA, calE, J=None):
        """``calA``, ``calE`` NV x NV, ``J`` NP x NV (or None), any scipy format."""
        self._zdev = None
        arp, aci, av, ash = as_csr(calA)
        erp, eci, ev, esh = as_csr(calE)
        nv = ash[0]
        if ash != (nv, nv) or esh != (nv, nv):
            raise ValueError("calA / calE must be square and of equal size")
        if J is not None and J.shape[0] > 0:
            jrp, jci, jv, jsh = as_csr(J)
            if jsh[1] != nv:
                raise ValueError("J has the wrong number of columns")
            np_ = jsh[0]
            _chk(self._lib.ricadi_set_operator(self._h, nv, np_, _i(arp), _i(aci), _d(av),
                                               _i(erp), _i(eci), _d(ev), _i(jrp), _i(jci),
                                               _d(jv)))
        else:
            np_ = 0
            _chk(self._lib.ricadi_set_operator(self._h, nv, 0, _i(arp), _i(aci), _d(av),
                                               _i(erp), _i(eci), _d(ev), None, None, None))
        self.nv, self.np_ = nv, np_
        self.n = nv + np_

    def clear_cache(self):
        _chk(self._lib.ricadi_clear_cache(self._h))

    def set_dims(self, nv):
        """Dimension-only context: enough for compress() and gain(MT=...)."""
        self._zdev = None
        _chk(self._lib.ricadi_set_dims(self._h, int(nv)))
        self.nv, self.np_, self.n = int(nv), 0, int(nv)

    def set_lowrank(self, U=None, V=None):
        """Operator becomes ``beta*A + alpha*E - U V^T`` (both NV x q)."""
        if U is None or V is None:
            _chk(self._lib.ricadi_set_lowrank(self._h, None, None, 0))
            return
        U = as_panel(U, self.nv)
        V = as_panel(V, self.nv)
        if U.shape != V.shape:
            raise ValueError("U and V must have the same shape")
        _chk(self._lib.ricadi_set_lowrank(self._h, _d(U), _d(V), U.shape[1]))

    def set_recycle(self, depth):
        """Depth of the recycling ring for DIRECT solve calls (the ADI drivers use their own, 3)."""
        _chk(self._lib.ricadi_set_recycle(self._h, int(depth)))
        self.recycle_depth = int(depth)

    def set_exchange(self, group=None, panel_cols=MAX_M, per_rank=2, transport=None):
        """Shard the ADI sweeps of this context over the ranks of a ``torch.distributed`` process group
        (SURVEY.md 8e): ONE all-gather of the solution panels per sweep, ``per_rank`` panels of
        ``n x panel_cols`` per rank.  ``transport``:

        * ``"rccl"`` (default for groups of the nccl backend): the library joins an RCCL communicator of its
          own -- rank 0's ``ncclGetUniqueId`` bytes go round through the group -- and enqueues
          ``ncclAllGather`` on its stream between the solves and the recombination (``ricadi_set_exchange_rccl``):
          no host synchronisation, no Python in the sweep;
        * ``"callback"`` (gloo groups: CPU tests, several ranks on one GPU): the library calls back for
          ``all_gather_into_tensor`` on two device buffers allocated here.

        ``group=False`` removes the exchange."""
        import torch
        import torch.distributed as dist
        if group is False or not (dist.is_available() and dist.is_initialized()):
            _chk(self._lib.ricadi_set_exchange(self._h, 0, 1, None, None, None, None, 0))
            self._xchg = None
            return
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        if transport is None:
            transport = "rccl" if dist.get_backend(group) == "nccl" else "callback"
        if world == 1 and transport != "rccl":
            _chk(self._lib.ricadi_set_exchange(self._h, 0, 1, None, None, None, None, 0))
            self._xchg = None
            return
        count = int(per_rank) * self.n * int(panel_cols) + 512        # + 4096 bytes of control messages
        if transport == "rccl":
            cur = getattr(self, "_xchg", None)
            if cur is not None and cur[0] == "rccl" and cur[3] is group:
                # same communicator, larger buffers (wider panels)
                _chk(self._lib.ricadi_set_exchange_rccl(self._h, rank, world, None, None, count * 8))
                self._xchg = ("rccl", None, None, group, count)
                return
            ident = [None]
            if rank == 0:
                buf = C.create_string_buffer(128)
                _chk(self._lib.ricadi_rccl_unique_id(buf, 128))
                ident[0] = buf.raw
            if world > 1:
                dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0,
                                           group=group)
            torch.cuda.synchronize()
            rc = self._lib.ricadi_set_exchange_rccl(self._h, rank, world, ident[0], None, count * 8)
            ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device="cuda")
            if world > 1:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)       # all ranks take the same transport
            if int(ok.item()) == 1:
                self._xchg = ("rccl", None, None, group, count)
                return
            import warnings
            warnings.warn("ricadi: the library's own RCCL communicator could not be set up ({0}); the all-gather of the "
                          "sharded sweeps goes through torch.distributed instead".format(
                              self._lib.ricadi_last_error().decode() if rc != 0 else "another rank failed"),
                          RuntimeWarning)
            _chk(self._lib.ricadi_set_exchange(self._h, 0, 1, None, None, None, None, 0))
            if world == 1:
                self._xchg = None
                return
        dev = torch.device("cuda", torch.cuda.current_device())
        send = torch.zeros(count, dtype=torch.float64, device=dev)
        recv = torch.zeros(count * world, dtype=torch.float64, device=dev)

        def gather(user, sptr, rptr, nbytes):
            try:
                k = int(nbytes) // 8
                so = (int(sptr) - send.data_ptr()) // 8           # the library also calls with pointers INTO the
                ro = (int(rptr) - recv.data_ptr()) // 8           # buffers (control messages in their tails)
                dist.all_gather_into_tensor(recv[ro:ro + k * world], send[so:so + k], group=group)
                torch.cuda.current_stream().synchronize()
                return 0
            except Exception as e:                        # never unwind through the C frames
                import sys
                print("ricadi exchange callback: {0!r}".format(e), file=sys.stderr, flush=True)
                return 1

        cb = EXCHANGE_FN(gather)
        torch.cuda.synchronize()
        _chk(self._lib.ricadi_set_exchange(self._h, rank, world, C.cast(cb, _vp), None, send.data_ptr(),
                                           recv.data_ptr(), count * 8))
        self._xchg = (cb, send, recv, group, count)          # keep the callback and the buffers alive

    def exchange_count(self):
        """Collectives this context has issued so far (one per sharded ADI sweep + control messages)."""
        k = C.c_int64(0)
        _chk(self._lib.ricadi_exchange_count(self._h, C.byref(k)))
        return int(k.value)

    # -- kernels ----------------------------------------------------------
    def _need_op(self):
        if not self.n:
            raise RuntimeError("ricadi: no operator set on this context")

    def spmm(self, alpha, beta, X):
        self._need_op()
        X = as_panel(X, self.n)
        Y = np.empty_like(X)
        _chk(self._lib.ricadi_spmm(self._h, alpha, beta, _d(X), X.shape[1], _d(Y)))
        return Y

    def precond_apply(self, alpha, beta, R):
        self._need_op()
        R = as_panel(R, self.n)
        Z = np.empty_like(R)
        _chk(self._lib.ricadi_precond_apply(self._h, alpha, beta, _d(R), R.shape[1], _d(Z)))
        return Z

    def shift_solve(self, alpha, beta, R, Rp=None, strict=True):
        """Solve ``S(alpha,beta) [V;L] = [R;Rp]``; wide panels go in chunks."""
        self._need_op()
        R = as_panel(R, self.nv)
        m = R.shape[1]
        Rp = None if Rp is None else as_panel(Rp, self.np_)
        X = np.empty((self.n, m))
        iters, relres = 0, np.zeros(m)
        for c0 in range(0, m, MAX_M):
            c1 = min(m, c0 + MAX_M)
            Rc = np.ascontiguousarray(R[:, c0:c1])
            Rpc = None if Rp is None else np.ascontiguousarray(Rp[:, c0:c1])
            Xc = np.empty((self.n, c1 - c0))
            it = C.c_int(0)
            rr = np.zeros(c1 - c0)
            rc = self._lib.ricadi_shift_solve(self._h, alpha, beta, _d(Rc),
                                              None if Rpc is None else _d(Rpc), c1 - c0,
                                              _d(Xc), C.byref(it), _d(rr))
            _chk(rc, allow_noconv=not strict)
            X[:, c0:c1] = Xc
            iters += it.value
            relres[c0:c1] = rr
        return X, iters, relres

    # -- solvers ----------------------------------------------------------
    def lyap_adi(self, shifts, W, prm, fetch=True):
        W = as_panel(W, self.nv)
        m = W.shape[1]
        sh = np.ascontiguousarray(shifts, dtype=np.float64)
        cap = prm.adi_max_steps * m
        Z = np.empty((self.nv, cap)) if fetch else None
        cc = C.c_int(0)
        stats = np.zeros(8)
        self._zdev = None
        _chk(self._lib.ricadi_lyap_adi(self._h, _d(sh), sh.size, _d(W), m, C.byref(prm),
                                       None if Z is None else _d(Z), C.byref(cc), _d(stats)))
        c = cc.value
        if fetch:
            Z = Z.ravel()[:self.nv * c].reshape(self.nv, c)
        info = dict(adi_steps=int(stats[0]), adi_rel_newZ=stats[1], gmres_iters=int(stats[2]),
                    shift_solves=int(stats[3]), res_fro=stats[4], cols=c,
                    gmres_nonconverged=int(stats[5]), gmres_worst_relres=stats[6],
                    storage_escalations=int(stats[7]))
        _warn_nonconverged(info)
        return Z, info

    def ric_newtonadi(self, shifts, B, W, prm, Z0=None, oldB=None, fetch=True):
        B = as_panel(B, self.nv)
        W = as_panel(W, self.nv)
        sh = np.ascontiguousarray(shifts, dtype=np.float64)
        nb, mw = B.shape[1], W.shape[1]
        Z0 = None if Z0 is None else as_panel(Z0, self.nv)
        oldB = None if oldB is None else as_panel(oldB, self.nv)
        cap = prm.adi_max_steps * (mw + nb)
        Z = np.empty((self.nv, cap)) if fetch else None
        cc = C.c_int(0)
        stats = np.zeros(12)
        self._zdev = None
        _chk(self._lib.ricadi_ric_newtonadi(
            self._h, _d(sh), sh.size, _d(B), nb, _d(W), mw,
            None if Z0 is None else _d(Z0), 0 if Z0 is None else Z0.shape[1],
            None if oldB is None else _d(oldB), C.byref(prm),
            None if Z is None else _d(Z), cap, C.byref(cc), _d(stats)))
        c = cc.value
        if fetch:
            Z = Z.ravel()[:self.nv * c].reshape(self.nv, c)
            # The factor also stays on the device.  The returned array is made read-only, so that a later
            # gain(B, Z=<this very array>) may use the device copy instead of uploading it again
            # (get_mTzzTtb right after the Newton iteration: optcont_main.py:488-506).
            Z.flags.writeable = False
            self._zdev = Z
        info = dict(nwtn_steps=int(stats[0]), upd_abs=stats[1], upd_rel=stats[2],
                    adi_steps=int(stats[3]), gmres_iters=int(stats[4]),
                    shift_solves=int(stats[5]), cols=c,
                    gmres_nonconverged=int(stats[6]), gmres_worst_relres=stats[7],
                    lyap_res_fro=stats[8], lyap_rhs_fro=stats[9], storage_escalations=int(stats[10]),
                    adi_sweeps=int(stats[11]))
        _warn_nonconverged(info)
        return Z, info

    def factor_get_dev(self, z_ptr, c):
        """Copy the context's resident factor (NV x c, packed) into a device buffer."""
        _chk(self._lib.ricadi_factor_get_dev(self._h, z_ptr, int(c)))

    def ric_newtonadi_dev(self, shifts, B_t, W_t, prm, Z0_t=None, old_t=None):
        """``ricadi_ric_newtonadi_dev``: every panel is a torch CUDA tensor (float64, contiguous, NV rows);
        the new iterate comes back as a fresh NV x c CUDA tensor.  No PCIe traffic."""
        import torch
        sh = np.ascontiguousarray(shifts, dtype=np.float64)
        for t in (B_t, W_t, Z0_t, old_t):
            if t is not None and not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
                                      and t.dim() == 2 and t.shape[0] == self.nv):
                raise ValueError("device panels must be contiguous float64 CUDA tensors with NV rows")
        cc = C.c_int(0)
        stats = np.zeros(12)
        self._zdev = None
        torch.cuda.current_stream().synchronize()          # the library runs on its own stream
        _chk(self._lib.ricadi_ric_newtonadi_dev(
            self._h, _d(sh), sh.size, B_t.data_ptr(), B_t.shape[1], W_t.data_ptr(), W_t.shape[1],
            None if Z0_t is None else Z0_t.data_ptr(), 0 if Z0_t is None else Z0_t.shape[1],
            None if old_t is None else old_t.data_ptr(), C.byref(prm), C.byref(cc), _d(stats)))
        c = cc.value
        Zt = torch.empty((self.nv, c), dtype=torch.float64, device=B_t.device)
        if c > 0:
            _chk(self._lib.ricadi_factor_get_dev(self._h, Zt.data_ptr(), c))
        info = dict(nwtn_steps=int(stats[0]), upd_abs=stats[1], upd_rel=stats[2],
                    adi_steps=int(stats[3]), gmres_iters=int(stats[4]),
                    shift_solves=int(stats[5]), cols=c,
                    gmres_nonconverged=int(stats[6]), gmres_worst_relres=stats[7],
                    lyap_res_fro=stats[8], lyap_rhs_fro=stats[9], storage_escalations=int(stats[10]),
                    adi_sweeps=int(stats[11]))
        _warn_nonconverged(info)
        return Zt, info

    def compress(self, Z=None, thresh=None, k=None):
        """``Z=None`` compresses the factor left on the device."""
        self._zdev = None
        if Z is not None:
            Z = as_panel(Z, self.nv)
            c = Z.shape[1]
        else:
            cc = C.c_int(0)
            _chk(self._lib.ricadi_factor_cols(self._h, C.byref(cc)))
            c = cc.value
        out = np.empty((self.nv, max(c, 1)))
        kk = C.c_int(0)
        sv = np.zeros(max(c, 1))
        _chk(self._lib.ricadi_compress(self._h, None if Z is None else _d(Z), c,
                                       -1.0 if thresh is None else float(thresh),
                                       0 if k is None else int(k), _d(out), C.byref(kk), _d(sv)))
        kk = kk.value
        return out.ravel()[:self.nv * kk].reshape(self.nv, kk).copy(), sv[:min(c, self.nv)]

    def recompress(self, Z, rel=0.0):
        """The drivers' internal recompression (pivoted Cholesky of the Gram matrix, no eigensolver):
        ``Zc`` with ``Zc Zc^T = Z Z^T`` up to ``rel^2 ||Z Z^T||`` (``rel=0``: the drivers' own level)."""
        Z = as_panel(Z, self.nv)
        c = Z.shape[1]
        out = np.empty((self.nv, c))
        kk = C.c_int(0)
        _chk(self._lib.ricadi_recompress(self._h, _d(Z), c, float(rel), _d(out), C.byref(kk)))
        kk = kk.value
        return out.ravel()[:self.nv * kk].reshape(self.nv, kk).copy()

    def gain(self, B, Z=None, MT=None):
        """``MT (Z (Z^T B))`` with ``MT`` = calE of the context when None."""
        B = as_panel(B, self.nv)
        K = np.empty_like(B)
        if Z is not None and Z is getattr(self, "_zdev", None) and not Z.flags.writeable:
            Z = None          # the device-resident factor IS this array (read-only since it was returned)
        if Z is not None:
            Z = as_panel(Z, self.nv)
        c = 0 if Z is None else Z.shape[1]
        if MT is not None:
            rp, ci, v, sh = as_csr(MT)
            _chk(self._lib.ricadi_gain(self._h, _i(rp), _i(ci), _d(v),
                                       None if Z is None else _d(Z), c, _d(B), B.shape[1], _d(K)))
        else:
            _chk(self._lib.ricadi_gain(self._h, None, None, None,
                                       None if Z is None else _d(Z), c, _d(B), B.shape[1], _d(K)))
        return K

    def lyap_res_norm(self, Z, W):
        Z = as_panel(Z, self.nv)
        W = as_panel(W, self.nv)
        out = C.c_double(0.0)
        _chk(self._lib.ricadi_lyap_res_norm(self._h, _d(Z), Z.shape[1], _d(W), W.shape[1],
                                            C.byref(out)))
        return out.value

    def factor_get(self):
        cc = C.c_int(0)
        _chk(self._lib.ricadi_factor_cols(self._h, C.byref(cc)))
        Z = np.empty((self.nv, cc.value))
        _chk(self._lib.ricadi_factor_get(self._h, _d(Z), cc.value))
        return Z

    # -- device-pointer level (torch tensors' data_ptr()) ------------------
    def spmm_dev(self, alpha, beta, x_ptr, m, y_ptr):
        _chk(self._lib.ricadi_spmm_dev(self._h, alpha, beta, x_ptr, m, y_ptr))

    def shift_solve_dev(self, alpha, beta, r_ptr, m, x_ptr, strict=True):
        it = C.c_int(0)
        rr = np.zeros(m)
        rc = self._lib.ricadi_shift_solve_dev(self._h, alpha, beta, r_ptr, m, x_ptr,
                                              C.byref(it), _d(rr))
        _chk(rc, allow_noconv=not strict)
        return it.value, rr

    def shift_solve_batch_dev(self, alphas, betas, r_ptr, r_stride, m, x_ptr, strict=True):
        """One batched solve for the shifts ``(alphas[g], betas[g])``; the right-hand sides
        are ``r_ptr + g*r_stride`` (``r_stride = 0``: shared), the solutions the ``n x m``
        panels ``x_ptr + g*n*m``.  Returns ``(iters per group, relres (ng x m))``."""
        al = np.ascontiguousarray(alphas, dtype=np.float64)
        be = np.ascontiguousarray(betas, dtype=np.float64)
        ng = al.size
        its = (C.c_int * ng)()
        rr = np.zeros((ng, m))
        rc = self._lib.ricadi_shift_solve_batch_dev(self._h, ng, _d(al), _d(be), r_ptr,
                                                    int(r_stride), m, x_ptr, its, _d(rr))
        _chk(rc, allow_noconv=not strict)
        return list(its), rr

    def time_spmm_batch_dev(self, alphas, betas, x_ptr, m, y_ptr, reps):
        """Milliseconds per batched saddle-SpMM launch (ng panels, one shift each)."""
        al = np.ascontiguousarray(alphas, dtype=np.float64)
        be = np.ascontiguousarray(betas, dtype=np.float64)
        ms = C.c_double(0.0)
        _chk(self._lib.ricadi_time_spmm_batch_dev(self._h, al.size, _d(al), _d(be), x_ptr, m, y_ptr,
                                                  reps, C.byref(ms)))
        return ms.value

    TK = dict(spmm=0, block_v=1, block_p=2, coarse=3, spmm_sy=4, dots=5, update_dots=6, update=7,
              precond=8, restrict=9, pc_restrict=10, pc_coarse=11, pc_sy_prows=12, pc_two_term=13,
              pc_jprod=14, pc_schur=15, pc_rect=16)

    def time_kernel_dev(self, which, alphas, betas, m, nvec=7, reps=100):
        """Milliseconds per launch of one hot-path kernel class (``Context.TK``) as the
        batched GMRES issues it for ``len(alphas)`` groups of width ``m``."""
        al = np.ascontiguousarray(alphas, dtype=np.float64)
        be = np.ascontiguousarray(betas, dtype=np.float64)
        ms = C.c_double(0.0)
        _chk(self._lib.ricadi_time_kernel_dev(self._h, int(self.TK.get(which, which)), al.size, _d(al),
                                              _d(be), int(m), int(nvec), int(reps), C.byref(ms)))
        return ms.value

    def setup_info(self):
        a = (C.c_int * 21)()
        _chk(self._lib.ricadi_setup_info(self._h, a, 21))
        return dict(zip(("nv", "np", "nbv", "nbp", "bs", "kc", "spmm_row_blocks", "spmm_max_cols", "levels",
                         "dense_coarse", "fp16_vector_input", "rect_ks", "two_term_ks", "np_", "nnz_j",
                         "nnz_sy", "nnz_restriction", "coarse_route", "k1_variant", "fp32_intermediate", "fp32_operator_output"),
                        list(a)))

    def dense_inverse_batch(self, mats):
        """In-place inverses of a batch of dense matrices by the setup's coarse-matrix routine; returns
        (inverses, route) -- route 0 block Gauss-Jordan, 1 rocSOLVER with partial pivoting."""
        A = np.ascontiguousarray(mats, dtype=np.float64).copy()
        if A.ndim != 3 or A.shape[1] != A.shape[2]:
            raise ValueError("mats must be nb x k x k")
        route = C.c_int(-1)
        _chk(self._lib.ricadi_dense_inverse_batch(self._h, A.shape[1], A.shape[0], _d(A), C.byref(route)))
        return A, int(route.value)

    def time_qr_dev(self, z_ptr, c, reps):
        ms = C.c_double(0.0)
        _chk(self._lib.ricadi_time_qr_dev(self._h, z_ptr, int(c), int(reps), C.byref(ms)))
        return ms.value

    def sweep_recombine_dev(self, G, u_ptr, m, rinv, cinv1, z_ptr, w_ptr):
        """Cauchy recombination of one sweep on the device; returns ``||Z-block||_F^2``."""
        ri = np.ascontiguousarray(rinv, dtype=np.float64)
        ci = np.ascontiguousarray(cinv1, dtype=np.float64)
        n2 = C.c_double(0.0)
        _chk(self._lib.ricadi_sweep_recombine_dev(self._h, int(G), u_ptr, m, _d(ri), _d(ci), z_ptr,
                                                  w_ptr, C.byref(n2)))
        return n2.value

    def sweep_recombine_slots_dev(self, nslot, G, u_ptr, m, coefz, coefw, z_ptr, w_ptr):
        """Recombination of one sweep from ``nslot`` gathered panels (see the header)."""
        cz = np.ascontiguousarray(coefz, dtype=np.float64)
        cw = np.ascontiguousarray(coefw, dtype=np.float64)
        if cz.shape != (nslot, G) or cw.shape != (nslot,):
            raise ValueError("coefficient tables do not match (nslot, G)")
        n2 = C.c_double(0.0)
        bn = np.zeros(int(G))
        _chk(self._lib.ricadi_sweep_recombine_slots_dev(self._h, int(nslot), int(G), u_ptr, m, _d(cz),
                                                        _d(cw), z_ptr, w_ptr, C.byref(n2), _d(bn)))
        return n2.value, bn

    def apply_e_dev(self, coef, v_ptr, m, w_ptr):
        _chk(self._lib.ricadi_apply_e_dev(self._h, coef, v_ptr, m, w_ptr))

    def lincomb_dev(self, nrows, m, coef, basis_ptr, stride, out_ptr):
        cf = np.ascontiguousarray(coef, dtype=np.float64)
        _chk(self._lib.ricadi_lincomb_dev(self._h, nrows, m, cf.size, basis_ptr, int(stride),
                                          _d(cf), out_ptr))

    def gain_dev(self, coef, z_ptr, c, ldz, b_ptr, nb, k_ptr):
        _chk(self._lib.ricadi_gain_dev(self._h, coef, z_ptr, c, ldz, b_ptr, nb, k_ptr))

    def panel_norms_dev(self, w_ptr, nrows, m):
        g = C.c_double(0.0)
        t = C.c_double(0.0)
        _chk(self._lib.ricadi_panel_norms_dev(self._h, w_ptr, nrows, m, C.byref(g), C.byref(t)))
        return g.value, t.value

    def time_spmm_dev(self, alpha, beta, x_ptr, m, y_ptr, reps):
        ms = C.c_double(0.0)
        _chk(self._lib.ricadi_time_spmm_dev(self._h, alpha, beta, x_ptr, m, y_ptr, reps,
                                            C.byref(ms)))
        return ms.value


def _time_gram_dev(self, z_ptr, c, g_ptr, reps):
    ms = C.c_double(0.0)
    _chk(self._lib.ricadi_time_gram_dev(self._h, z_ptr, c, g_ptr, reps, C.byref(ms)))
    return ms.value


Context.time_gram_dev = _time_gram_dev


def _qr(self, Z, want_q=True):
    """Thin QR by TSQR panels + block Gram-Schmidt (K5); returns (Q or None, R)."""
    Z = as_panel(Z, self.nv)
    c = Z.shape[1]
    R = np.empty((c, c))
    Q = np.empty_like(Z) if want_q else None
    _chk(self._lib.ricadi_qr(self._h, _d(Z), c, None if Q is None else _d(Q), _d(R)))
    return Q, R


Context.qr = _qr


def host_aggregate(pattern, bsize):
    """Greedy BFS aggregation (host logic of the preconditioner setup)."""
    rp, ci, _, sh = as_csr(pattern)
    blk = np.empty(sh[0], dtype=np.int32)
    nb = load().ricadi_host_aggregate(sh[0], _i(rp), _i(ci), int(bsize), _i(blk))
    if nb < 0:
        _chk(nb)
    return blk, nb


def host_deal(shifts, world):
    """Owner rank of every shift of an ADI shift list (``ricadi_host_deal``)."""
    sh = np.ascontiguousarray(shifts, dtype=np.float64)
    owner = np.empty(sh.size, dtype=np.int32)
    _chk(load().ricadi_host_deal(_d(sh), sh.size, int(world), _i(owner)))
    return owner


def host_cauchy(shifts):
    """``(R^-1, C^-1 1)`` of the Cauchy matrix of a shift sweep."""
    sh = np.ascontiguousarray(shifts, dtype=np.float64)
    g = sh.size
    rinv = np.empty((g, g))
    c1 = np.empty(g)
    _chk(load().ricadi_host_cauchy(_d(sh), g, _d(rinv), _d(c1)))
    return rinv, c1


def host_plan_levels(calA, calE, J, **opts):
    """The preconditioner hierarchy ``set_operator`` would choose (``ricadi_host_plan_levels``; host only):
    dict(levels, kc, kcv, kcp, smoothed)."""
    a, e, j = as_csr(calA), as_csr(calE), as_csr(J)
    o = default_opts(**opts)
    out = np.zeros(5, dtype=np.int32)
    _chk(load().ricadi_host_plan_levels(a[3][0], j[3][0], _i(a[0]), _i(a[1]), _d(a[2]), _i(e[0]), _i(e[1]), _d(e[2]),
                                        _i(j[0]), _i(j[1]), _d(j[2]), C.byref(o), _i(out)))
    return dict(levels=int(out[0]), kc=int(out[1]), kcv=int(out[2]), kcp=int(out[3]), smoothed=bool(out[4]))


def host_sa_criterion(calA):
    """``(on, rowsum_ratio, skew_ratio)``: whether the setup would smooth the velocity aggregates for ``calA``
    (``ricadi_host_sa_criterion``; host only)."""
    rp, ci, v, sh = as_csr(calA)
    rs, sk, on = C.c_double(0.0), C.c_double(0.0), C.c_int(0)
    _chk(load().ricadi_host_sa_criterion(sh[0], _i(rp), _i(ci), _d(v), C.byref(rs), C.byref(sk), C.byref(on)))
    return bool(on.value), rs.value, sk.value
