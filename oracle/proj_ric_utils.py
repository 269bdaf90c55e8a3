"""ORACLE -- test infrastructure, not product code.

CPU restatement (numpy / scipy SuperLU) of the low-rank Newton-ADI solver the
reference imports as ``sadptprj_riclyap_adi.proj_ric_utils`` (``pru``).  The
package is NOT part of ``/root/reference`` (only imported:
``optcont_main.py:14``, ``solve_dae_ric.py:4``,
``tests/test_units_compfacres_compress.py:4``) and is unpinned, so the
algorithm body is restated from the standard low-rank ADI / Newton-Kleinman
literature for index-2 DAEs and anchored on the
reference's call sites and its unit test.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.

PARITY UNPINNED: the reference holds no golden vectors for this path
(SURVEY.md section 8c).  Pinned instead: the equation and its sign
(``tests/test_units_compfacres_compress.py:75-79``), the projector
(``:70-73``), ``comp_proj_lyap_res_norm`` returning the *squared* norm
(``:82``), the gain formula (``optcont_main.py:505``), the default
tolerances (``optcont_main.py:122-135``).
"""
from __future__ import annotations

import time

import numpy as np
import scipy.sparse as sps

from . import lin_alg_utils as lau

__all__ = [
    "solve_proj_lyap_stein", "proj_alg_ric_newtonadi", "compress_Zsvd",
    "get_mTzzTtb", "comp_proj_lyap_res_norm", "comp_diff_zzt_fnorm",
    "DEFAULT_MS",
]

# Built-in shift list used when ``adi_dict`` has no ``'ms'`` key
# (tests/test_units_compfacres_compress.py:54-64 relies on such a default).
# The upstream values are not in the container [INFERRED].
DEFAULT_MS = [-30.0, -20.0, -10.0, -5.0, -3.0, -1.0]


def _dense(a):
    if sps.issparse(a):
        return np.asarray(a.todense())
    a = np.asarray(a, dtype=float)
    return a.reshape(-1, 1) if a.ndim == 1 else a


def get_mTzzTtb(MT, Z, tb, output=None):
    """``MT * (Z * (Z^T * tb))`` -- the feedback gain up to sign.

    ``K = -M^T Z Z^T B`` at ``optcont_main.py:505``, ``solve_dae_ric.py:101,189``;
    affine term at ``solve_dae_ric.py:183``.  ``tb`` sparse or dense.
    """
    ztb = Z.T @ tb if sps.issparse(tb) else np.dot(Z.T, _dense(tb))
    ztb = _dense(ztb)
    return MT @ np.dot(Z, ztb)


def comp_diff_zzt_fnorm(Z1, Z0):
    """``||Z1 Z1^T - Z0 Z0^T||_F`` without cancellation (QR of ``[Z1, Z0]``)."""
    if Z0 is None or Z0.shape[1] == 0:
        G = Z1.T @ Z1
        return np.linalg.norm(G)
    ZZ = np.hstack([Z1, Z0])
    R = np.linalg.qr(ZZ, mode="r")
    sgn = np.r_[np.ones(Z1.shape[1]), -np.ones(Z0.shape[1])]
    return np.linalg.norm((R * sgn) @ R.T)


class _ShiftedSolver:
    """Caches one saddle LU per distinct shift (what the reference amortises)."""

    def __init__(self, cala, cale, jmat, U=None, Vt=None, stats=None):
        self.cala, self.cale, self.jmat = cala, cale, jmat
        self.U = None if U is None else _dense(U)
        self.Vt = None if Vt is None else _dense(Vt)
        self.lus = {}
        self.stats = stats if stats is not None else {}
        self.stats.setdefault("lu_time", 0.0)
        self.stats.setdefault("solve_time", 0.0)
        self.stats.setdefault("n_lu", 0)
        self.stats.setdefault("n_shift_solves", 0)

    def solve(self, p, rhs):
        if p not in self.lus:
            t0 = time.perf_counter()
            self.lus[p] = lau.SaddleLU(self.cala + p * self.cale, self.jmat)
            self.stats["lu_time"] += time.perf_counter() - t0
            self.stats["n_lu"] += 1
        t0 = time.perf_counter()
        x = lau.solve_sadpnt_smw(rhsv=rhs, umat=self.U, vmat=self.Vt,
                                 sadlu=self.lus[p])[:self.cala.shape[0]]
        self.stats["solve_time"] += time.perf_counter() - t0
        self.stats["n_shift_solves"] += 1
        return x


def solve_proj_lyap_stein(amat=None, mmat=None, jmat=None, wmat=None,
                          umat=None, vmat=None, transposed=False,
                          adi_dict=None, nwtn_adi_dict=None, stats=None):
    """Low-rank ADI for the projected Lyapunov equation.

    With ``F = amat - umat*vmat``:
    ``F^T X M + M^T X F + W W^T = 0`` on the divergence-free space
    (sign pinned by ``tests/test_units_compfacres_compress.py:62-64,75-79``);
    ``transposed=True`` swaps the roles, ``F X M^T + M X F^T + W W^T = 0``.

    Residual-form LR-ADI with real shifts ``p_i < 0`` cycled from
    ``adi_dict['ms']`` (``run_optcont.py:18-19``), every solve through the
    saddle-point matrix ``[[cal_A + p cal_E, J^T], [J, 0]]``:

        W_0 = P^T W;  V_i = (cal_A + p_i cal_E)^-1_P W_{i-1};
        W_i = W_{i-1} - 2 p_i cal_E V_i;  Z_i = [Z_{i-1}, sqrt(-2 p_i) V_i]

    Stops when ``||Z_new||_F / ||Z_i||_F < adi_newZ_reltol`` or after
    ``adi_max_steps`` (``optcont_main.py:123-124``).  Returns a dict with
    ``'zfac'`` (``tests/...compress.py:64``) plus iteration bookkeeping.
    """
    adi_dict = nwtn_adi_dict if adi_dict is None else adi_dict
    adi_dict = {} if adi_dict is None else adi_dict
    ms = list(adi_dict.get("ms", DEFAULT_MS))
    max_steps = int(adi_dict.get("adi_max_steps", 200))
    reltol = float(adi_dict.get("adi_newZ_reltol", 1e-8))

    amat = sps.csr_matrix(amat)
    mmat = sps.csr_matrix(mmat)
    if transposed:
        cala, cale = amat, mmat
        U, Vt = umat, vmat
    else:
        cala, cale = amat.T.tocsr(), mmat.T.tocsr()
        # (amat - U V)^T = amat^T - V^T U^T
        U = None if vmat is None else _dense(vmat).T
        Vt = None if umat is None else _dense(umat).T
    W = _dense(wmat)
    # project the right-hand side factor once: W_0 = P^T W
    W = lau.app_prj_via_sadpnt(amat=cale.T.tocsr(), jmat=jmat, rhsv=W,
                               transposedprj=True) \
        if adi_dict.get("project_w", True) else W

    solver = _ShiftedSolver(cala, cale, jmat, U=U, Vt=Vt, stats=stats)
    blocks = []
    znorm2 = 0.0
    res_hist = [np.linalg.norm(W.T @ W)]
    rel = np.inf
    step = 0
    for step in range(1, max_steps + 1):
        p = float(ms[(step - 1) % len(ms)])
        V = solver.solve(p, W)
        W = W - 2.0 * p * (cale @ V)
        Znew = np.sqrt(-2.0 * p) * V
        blocks.append(Znew)
        n2 = np.linalg.norm(Znew) ** 2
        znorm2 += n2
        res_hist.append(np.linalg.norm(W.T @ W))
        rel = np.sqrt(n2 / znorm2)
        if adi_dict.get("verbose", False):
            print("ADI step {0:3d}: shift {1:9.3e} rel new Z {2:9.3e} res {3:9.3e}"
                  .format(step, p, rel, res_hist[-1]))
        if rel < reltol:
            break
    Z = np.hstack(blocks) if blocks else np.zeros((W.shape[0], 0))
    return dict(zfac=Z, adi_steps=step, adi_rel_newZ=rel, res_hist=res_hist,
                resfac=W)


def proj_alg_ric_newtonadi(mmat=None, amat=None, jmat=None, bmat=None,
                           wmat=None, z0=None, mtxoldb=None,
                           transposed=False, nwtn_adi_dict=None, stats=None,
                           **kw):
    """Newton-Kleinman / low-rank ADI for the projected algebraic Riccati eq.

    ``cal_A X cal_E^T + cal_E X cal_A^T - cal_E X B B^T X cal_E^T + W W^T = 0``
    with ``cal_A = amat^T, cal_E = mmat^T`` (``transposed=True``: ``amat``,
    ``mmat`` as given) -- call sites ``optcont_main.py:488-492`` and
    ``solve_dae_ric.py:152-159``.  ``mtxoldb`` is a low-rank feedback from
    earlier outer steps, ``cal_A + mtxoldb*bmat^T`` (sign from
    ``solve_dae_ric.py:151,181,192-194``).

    Each step solves the Lyapunov equation with ``cal_A_k = cal_A - K_k B^T``,
    ``K_k = cal_E Z_k Z_k^T B`` and right-hand side factor ``[W, K_k]``.
    Stops on the update norm (``nwtn_upd_reltol`` / ``nwtn_upd_abstol``) or
    after ``nwtn_max_steps`` (``optcont_main.py:125-127``).  Returns a dict
    with ``'zfac'``.
    """
    nd = {} if nwtn_adi_dict is None else nwtn_adi_dict
    max_steps = int(nd.get("nwtn_max_steps", 16))
    reltol = float(nd.get("nwtn_upd_reltol", 5e-8))
    abstol = float(nd.get("nwtn_upd_abstol", 1e-7))

    amat = sps.csr_matrix(amat)
    mmat = sps.csr_matrix(mmat)
    cale = mmat if transposed else mmat.T.tocsr()
    B = _dense(bmat)
    W = _dense(wmat)
    Zk = None if z0 is None else _dense(z0)
    old = None if mtxoldb is None else _dense(mtxoldb)

    upd_hist = []
    steps = 0
    for steps in range(1, max_steps + 1):
        if Zk is None:
            Kk = np.zeros((amat.shape[0], B.shape[1]))
        else:
            Kk = get_mTzzTtb(cale, Zk, B)
        Kall = Kk if old is None else Kk - old
        # closed loop: cal_A - Kall B^T ; as amat - umat*vmat in amat's orientation
        if transposed:
            um, vm = Kall, B.T
        else:
            um, vm = B, Kall.T
        rhs = np.hstack([W, Kk]) if Zk is not None else W
        out = solve_proj_lyap_stein(amat=amat, mmat=mmat, jmat=jmat, wmat=rhs,
                                    umat=um, vmat=vm, transposed=transposed,
                                    adi_dict=nd, stats=stats)
        Znew = out["zfac"]
        upd = comp_diff_zzt_fnorm(Znew, Zk)
        nrm = np.linalg.norm(Znew.T @ Znew)
        upd_hist.append((upd, upd / nrm if nrm > 0 else np.inf, out["adi_steps"]))
        if nd.get("verbose", False):
            print("Newton step {0:2d}: |upd| {1:9.3e} rel {2:9.3e} ({3} ADI steps, "
                  "{4} columns)".format(steps, upd, upd_hist[-1][1], out["adi_steps"],
                                        Znew.shape[1]))
        Zk = Znew
        if upd < abstol or upd < reltol * nrm:
            break
    return dict(zfac=Zk, nwtn_steps=steps, upd_hist=upd_hist)


def compress_Zsvd(Z, thresh=None, k=None, shplot=False):
    """Column compression ``Zc Zc^T ~ Z Z^T`` by QR + SVD of R.

    ``Z = Q R``, ``R = U S V^T``, ``Zc = Q U[:, :k'] S[:k']`` with
    ``k' = min(k, #{s_i > thresh})`` (absolute threshold).  Call sites:
    ``optcont_main.py:498-499``, ``solve_dae_ric.py:162-163``,
    ``tests/test_units_compfacres_compress.py:92``.  ``shplot`` is accepted and
    ignored (it plots singular values upstream).
    """
    Q, R = np.linalg.qr(Z, mode="reduced")
    U, s, _ = np.linalg.svd(R, full_matrices=False)
    kk = s.size
    if thresh is not None:
        kk = int(np.sum(s > thresh))
    if k is not None:
        kk = min(kk, int(k))
    return Q @ (U[:, :kk] * s[:kk])


def comp_proj_lyap_res_norm(Z, amat=None, mmat=None, wmat=None, jmat=None,
                            umat=None, vmat=None):
    """Squared Frobenius norm of the projected Lyapunov residual, from factors.

    ``|| P^T (F^T X M + M^T X F + W W^T) P ||_F^2`` with ``X = Z Z^T``
    (``tests/test_units_compfacres_compress.py:75-82,104``; positional use
    ``comp_proj_lyap_res_norm(Z, F, M, W, J)``).  Never forms an NV x NV matrix:
    with ``G = P^T F^T Z``, ``H = P^T M^T Z``, ``Wp = P^T W`` the residual is
    ``[G,H,Wp] S [G,H,Wp]^T`` and its norm follows from the small Gram matrix.
    """
    F = sps.csr_matrix(amat)
    M = sps.csr_matrix(mmat)
    FtZ = F.T @ Z
    if umat is not None and vmat is not None:
        FtZ = FtZ - _dense(vmat).T @ (_dense(umat).T @ Z)
    stack = np.hstack([FtZ, M.T @ Z, _dense(wmat)])
    # project all factors at once: one saddle LU with M
    stack = lau.app_prj_via_sadpnt(amat=M, jmat=jmat, rhsv=stack, transposedprj=True)
    c = Z.shape[1]
    G, H, Wp = stack[:, :c], stack[:, c:2 * c], stack[:, 2 * c:]
    Ufac = np.hstack([G, H, Wp])
    Gram = Ufac.T @ Ufac
    nw = Wp.shape[1]
    S = np.zeros((2 * c + nw, 2 * c + nw))
    S[:c, c:2 * c] = np.eye(c)
    S[c:2 * c, :c] = np.eye(c)
    S[2 * c:, 2 * c:] = np.eye(nw)
    SG = S @ Gram
    return float(np.trace(SG @ SG))
