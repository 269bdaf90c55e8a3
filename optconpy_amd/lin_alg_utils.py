"""Host-side mirror of ``sadptprj_riclyap_adi.lin_alg_utils`` on the HIP path.

Same names, keyword arguments and return shapes as the functions the reference
calls (``/root/reference/optcont_main.py:232-236,398,405-408,421,424-425,510-514``;
``/root/reference/solve_dae_ric.py:77,81,92,94,97,100,108,192-194``;
``/root/reference/tests/test_units_compfacres_compress.py:71``).  Every solve
with an NV-sized matrix runs on the GPU through ``libricadi_hip.so`` (block-
Jacobi/coarse-level preconditioned GMRES instead of the reference's SuperLU);
only the NU x NU / NY x NY square roots stay on the host, as plain numpy.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sps

from . import backend

__all__ = [
    "solve_sadpnt_smw", "app_prj_via_sadpnt", "apply_massinv",
    "apply_invsqrt_fromright", "apply_sqrt_fromright", "app_luinv_to_spmat",
    "mm_dnssps",
]

# below this size a "mass matrix" is one of the small input/output-space
# matrices (optcont_main.py:43: NU = NY = 4 per component)
_SMALL = 512


def _dense(a):
    if sps.issparse(a):
        return np.asarray(a.todense())
    a = np.asarray(a, dtype=float)
    return a.reshape(-1, 1) if a.ndim == 1 else a


def mm_dnssps(A, B):
    """``A * B`` for any mix of dense / sparse factors (``optcont_main.py:232-236``)."""
    if sps.issparse(A) or sps.issparse(B):
        return _dense(A @ B)
    return np.dot(A, B)


def solve_sadpnt_smw(amat=None, jmat=None, rhsv=None, jmatT=None, umat=None,
                     vmat=None, rhsp=None, **kw):
    """Solve ``[[amat - umat*vmat, J^T],[J, 0]] x = [rhsv; rhsp]`` on the GPU.

    The low-rank term is handled inside the library as the reference does, by
    Sherman-Morrison-Woodbury around solves with the plain saddle operator
    (``x = y + W (V^T y)``, ``W = S^-1 [U;0] (I - V^T S^-1 U)^-1``), followed by an
    FP64 check of the closed-loop residual and, should it miss the tolerance,
    one GMRES refinement on the closed-loop operator itself (``RICADI_SMW=0``
    keeps the term inside the Krylov operator from the start).
    Returns the ``(NV+NP) x q`` solution like the reference
    (callers slice ``[:NV]``: ``solve_dae_ric.py:192-194``,
    ``optcont_main.py:510-514``).
    """
    if jmat is None and jmatT is not None:
        jmat = sps.csr_matrix(jmatT).T
    amat = sps.csr_matrix(amat)
    nv = amat.shape[0]
    # cal E enters with coefficient alpha = 0: it only lends its graph to the preconditioner's aggregation
    # (backend.mass_hint); without a remembered mass matrix the operator's own pattern serves
    hint = backend.mass_hint(nv)
    if hint is not None:
        # the hint must belong to this problem: its pattern lies inside the operator's (sampled rows)
        h = sps.csr_matrix(hint)
        for r in np.linspace(0, nv - 1, 48).astype(int):
            if not set(h.indices[h.indptr[r]:h.indptr[r + 1]]) <= set(amat.indices[amat.indptr[r]:amat.indptr[r + 1]]):
                hint = None
                break
    ctx = backend.context_for(amat, hint if hint is not None else sps.csr_matrix((nv, nv)), jmat)
    if umat is not None and vmat is not None:
        ctx.set_lowrank(_dense(umat), _dense(vmat).T)
    else:
        ctx.set_lowrank(None, None)
    try:
        X, _, _ = ctx.shift_solve(0.0, 1.0, _dense(rhsv), None if rhsp is None else _dense(rhsp))
    finally:
        ctx.set_lowrank(None, None)
    return X


def app_prj_via_sadpnt(amat=None, jmat=None, rhsv=None, jmatT=None, umat=None,
                       vmat=None, transposedprj=False):
    """Discrete Leray projector through one saddle solve (``optcont_main.py:405-408``).

    ``P = I - M^-1 J^T (J M^-1 J^T)^-1 J`` with ``M = amat``
    (``tests/test_units_compfacres_compress.py:70-73``); ``transposedprj=True``
    returns ``P^T rhsv``.
    """
    rhsv = _dense(rhsv)
    amat = sps.csr_matrix(amat)
    nv = amat.shape[0]
    if transposedprj:
        x = solve_sadpnt_smw(amat=amat, jmat=jmat, jmatT=jmatT, rhsv=rhsv,
                             umat=umat, vmat=vmat)[:nv]
        return amat @ x
    return solve_sadpnt_smw(amat=amat, jmat=jmat, jmatT=jmatT, rhsv=amat @ rhsv,
                            umat=umat, vmat=vmat)[:nv]


def app_luinv_to_spmat(alu_solve, Z):
    """Apply a caller-supplied ``factorized`` handle column-wise, dense result.

    (``tests/test_units_compfacres_compress.py:71``.)  The handle is the
    caller's; nothing is solved here.
    """
    Zd = _dense(Z)
    out = np.zeros_like(Zd, dtype=float)
    for c in range(Zd.shape[1]):
        out[:, c] = alu_solve(Zd[:, c])
    return out


def apply_massinv(M, rhsa, output=None):
    """``M^-1 rhsa`` (``optcont_main.py:398``; ``solve_dae_ric.py:77,81,100,108``).

    NV-sized ``M``: GPU solve (no constraint block).  The small input / output
    space mass matrices are solved densely on the host.
    """
    M = sps.csr_matrix(M)
    rhs = _dense(rhsa)
    n = M.shape[0]
    if n <= _SMALL:
        out = np.linalg.solve(M.toarray(), rhs)
    else:
        ctx = backend.context_for(M, sps.csr_matrix((n, n)), None)
        ctx.set_lowrank(None, None)
        out, _, _ = ctx.shift_solve(0.0, 1.0, rhs)
    if output == "sparse":
        return sps.csr_matrix(out)
    return out


def _sym_funm(M, fun):
    Md = _dense(M)
    if Md.shape[0] > _SMALL:
        raise ValueError("matrix square roots are only provided for the small "
                         "input/output-space mass matrices")
    w, Q = np.linalg.eigh(0.5 * (Md + Md.T))
    return (Q * fun(w)) @ Q.T


def apply_invsqrt_fromright(M, rhsa, output=None):
    """``rhsa * M^(-1/2)`` (``optcont_main.py:421,424-425``; ``solve_dae_ric.py:92,97``)."""
    out = mm_dnssps(rhsa, _sym_funm(M, lambda w: 1.0 / np.sqrt(w)))
    if output == "sparse":
        return sps.csr_matrix(out)
    return out


def apply_sqrt_fromright(M, rhsa, output=None):
    """``rhsa * M^(1/2)`` (``solve_dae_ric.py:94``)."""
    out = mm_dnssps(rhsa, _sym_funm(M, np.sqrt))
    if output == "sparse":
        return sps.csr_matrix(out)
    return out
