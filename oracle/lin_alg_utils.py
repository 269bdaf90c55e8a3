"""ORACLE -- test infrastructure, not product code.

CPU restatement (numpy / scipy SuperLU) of the saddle-point linear algebra the
reference imports as ``sadptprj_riclyap_adi.lin_alg_utils`` (``lau``).  That
package is NOT part of ``/root/reference`` (it is only imported:
``optcont_main.py:13``, ``solve_dae_ric.py:3``) and no version is pinned, so
the functions below restate its *published behaviour* from the reference's call
sites and its one unit test.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module.

PARITY UNPINNED: the reference holds no golden vectors for this path
(SURVEY.md section 8c); the oracle is pinned by the algebraic identities of
``/root/reference/tests/test_units_compfacres_compress.py:70-106`` only.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spsla

__all__ = [
    "solve_sadpnt_smw", "app_prj_via_sadpnt", "apply_massinv",
    "apply_invsqrt_fromright", "apply_sqrt_fromright", "app_luinv_to_spmat",
    "mm_dnssps", "saddle_matrix", "SaddleLU",
]


def _dense(a):
    if sps.issparse(a):
        return np.asarray(a.todense())
    a = np.asarray(a, dtype=float)
    return a.reshape(-1, 1) if a.ndim == 1 else a


def mm_dnssps(A, B):
    """``A * B`` for any mix of dense / sparse factors, dense result.

    Call site: ``optcont_main.py:232-236``.
    """
    if sps.issparse(A) or sps.issparse(B):
        out = A @ B
        return _dense(out)
    return np.dot(A, B)


def saddle_matrix(amat, jmat, jmatT=None):
    """``[[amat, J^T], [J, 0]]`` as CSC (the matrix SuperLU factorises)."""
    jT = jmat.T if jmatT is None else jmatT
    NP = jmat.shape[0]
    return sps.bmat([[sps.csr_matrix(amat), jT], [jmat, sps.csr_matrix((NP, NP))]],
                    format="csc")


class SaddleLU:
    """One sparse LU of the saddle-point matrix, applied column by column.

    This is what the reference amortises per ADI shift (SURVEY.md section 8a,
    row a2): ``spsla.factorized`` as at
    ``tests/test_units_compfacres_compress.py:70``.
    """

    def __init__(self, amat, jmat, jmatT=None):
        self.NV = amat.shape[0]
        self.NP = jmat.shape[0]
        self.lu = spsla.factorized(saddle_matrix(amat, jmat, jmatT))

    def solve(self, rhsv, rhsp=None):
        rhsv = _dense(rhsv)
        q = rhsv.shape[1]
        rhs = np.zeros((self.NV + self.NP, q))
        rhs[:self.NV] = rhsv
        if rhsp is not None:
            rhs[self.NV:] = _dense(rhsp)
        out = np.empty_like(rhs)
        for c in range(q):
            out[:, c] = self.lu(rhs[:, c])
        return out


def solve_sadpnt_smw(amat=None, jmat=None, rhsv=None, jmatT=None, umat=None,
                     vmat=None, rhsp=None, sadlu=None, return_alu=False):
    """Solve ``[[amat - umat*vmat, J^T], [J, 0]] x = [rhsv; rhsp]``.

    Sherman-Morrison-Woodbury around one LU of the sparse saddle matrix.
    Sign convention pinned by ``solve_dae_ric.py:173,181,189,192-194``
    (``umat = tau*(-M^T X B)``, ``vmat = B^T`` gives the closed loop
    ``M^T + tau (A+N)^T + tau M^T X B B^T``) and ``optcont_main.py:510-514``.
    Returns the full ``(NV+NP) x q`` solution; callers slice ``[:NV]``.
    """
    if sadlu is None:
        sadlu = SaddleLU(amat, jmat, jmatT)
    x = sadlu.solve(rhsv, rhsp)
    if umat is not None and vmat is not None:
        NV = sadlu.NV
        U = _dense(umat)
        V = _dense(vmat)                       # NU x NV
        AinvU = sadlu.solve(U)                 # (n, NU)
        # (S - [U;0][V,0])^{-1} = S^-1 + S^-1 U (I - V S^-1 U)^-1 V S^-1
        cap = np.eye(U.shape[1]) - V @ AinvU[:NV]
        x = x + AinvU @ np.linalg.solve(cap, V @ x[:NV])
    if return_alu:
        return x, sadlu
    return x


def app_prj_via_sadpnt(amat=None, jmat=None, rhsv=None, jmatT=None,
                       umat=None, vmat=None, transposedprj=False):
    """Discrete Leray projector through one saddle solve with ``amat`` (= M).

    ``P = I - M^-1 J^T (J M^-1 J^T)^-1 J``
    (``tests/test_units_compfacres_compress.py:70-73``).
    ``transposedprj=True`` returns ``P^T rhsv`` (call site
    ``optcont_main.py:405-408``), else ``P rhsv``.
    """
    rhsv = _dense(rhsv)
    NV = amat.shape[0]
    if transposedprj:
        # [[M, J^T],[J,0]] [x; l] = [r; 0]  =>  x = P M^-1 r,  M x = P^T r
        x = solve_sadpnt_smw(amat=amat, jmat=jmat, jmatT=jmatT, rhsv=rhsv,
                             umat=umat, vmat=vmat)[:NV]
        return amat @ x
    x = solve_sadpnt_smw(amat=amat, jmat=jmat, jmatT=jmatT, rhsv=amat @ rhsv,
                         umat=umat, vmat=vmat)[:NV]
    return x


def app_luinv_to_spmat(alu_solve, Z):
    """Apply a ``factorized`` handle column-wise to a (sparse) matrix, dense out.

    Call site: ``tests/test_units_compfacres_compress.py:71``.
    """
    Zd = _dense(Z)
    out = np.zeros_like(Zd, dtype=float)
    for c in range(Zd.shape[1]):
        out[:, c] = alu_solve(Zd[:, c])
    return out


def apply_massinv(M, rhsa, output=None):
    """``M^-1 rhsa``; ``output='sparse'`` returns a csr matrix.

    Call sites: ``optcont_main.py:398``; ``solve_dae_ric.py:77,81,100,108``.
    """
    lu = spsla.factorized(sps.csc_matrix(M))
    out = app_luinv_to_spmat(lu, rhsa)
    if output == "sparse":
        return sps.csr_matrix(out)
    return out


def _sym_funm(M, fun):
    Md = _dense(M)
    w, Q = np.linalg.eigh(0.5 * (Md + Md.T))
    return (Q * fun(w)) @ Q.T


def apply_invsqrt_fromright(M, rhsa, output=None):
    """``rhsa * M^(-1/2)`` for a small s.p.d. ``M``.

    Call sites: ``optcont_main.py:421,424-425``; ``solve_dae_ric.py:92,97``.
    """
    out = mm_dnssps(rhsa, _sym_funm(M, lambda w: 1.0 / np.sqrt(w)))
    if output == "sparse":
        return sps.csr_matrix(out)
    return out


def apply_sqrt_fromright(M, rhsa, output=None):
    """``rhsa * M^(1/2)`` for a small s.p.d. ``M`` (``solve_dae_ric.py:94``)."""
    out = mm_dnssps(rhsa, _sym_funm(M, np.sqrt))
    if output == "sparse":
        return sps.csr_matrix(out)
    return out
