"""Closed-loop forward simulation with the computed feedback, and the cost functional.

Counterpart of what ``optcon_nse`` does after the Riccati solve
(``/root/reference/optcont_main.py:529-536,609-626``: ``snu.solve_nse(closed_loop=True,
feedbackthroughdict=..., tb_mat=...)``) and of ``eval_costfunc``
(``optcont_main.py:213-264``).  ``snu`` (dolfin_navier_scipy) is not available,
so the time stepper here is this repo's own implicit Euler scheme for the
*linearised* flow (the reference's ``linearized_nse=True`` branch):

    M (v_{k+1} - v_k)/tau + (A + N(t_{k+1})) v_{k+1} + J^T p = f + B~ u_{k+1},
    u = mtxtb(t)^T v + B~^T w(t)          (mtxtb = -M^T Z Z^T B~ is what the Riccati sweep stores)
    J v_{k+1} = 0

the control the reference applies (``optcont_main.py:218``: "u is a R.-1B(XMv+w)" with
``X = -Z Z^T``; evaluated as ``mtxtb^T v + tb^T w`` at ``optcont_main.py:244-248``).  Every step is one saddle-point
solve with the low-rank term ``- B~ mtxtb^T`` inside the operator --
``lau.solve_sadpnt_smw`` on the GPU (``solve_dae_ric.py:192-194`` uses the same call).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sps

from . import lin_alg_utils as _lau

from . import proj_ric_utils as _pru

__all__ = ["simulate_linearized_flow", "eval_costfunc", "steady_state_feedback"]


def steady_state_feedback(mmat=None, amat=None, jmat=None, convc_mat=None, tb_mat=None,
                          trct_mat=None, mc_mat=None, fv=None, ystar0=None, nwtn_adi_dict=None,
                          z0=None, comprz_thresh=None, comprz_maxc=None, store=None,
                          datastr="stst", pru=None, lau=None):
    """The steady-state branch of ``optcon_nse`` (``optcont_main.py:488-521``), call for call:

        Z      = pru.proj_alg_ric_newtonadi(mmat=M, amat=-A-N, jmat=J, bmat=B~, wmat=C~^T, z0)['zfac']
        Z      = pru.compress_Zsvd(Z, thresh, k)                         (if asked for)
        mtxtb  = -pru.get_mTzzTtb(M^T, Z, B~);   mtxfv = -pru.get_mTzzTtb(M^T, Z, fv)
        w      = lau.solve_sadpnt_smw(amat=A^T+N^T, jmat=J, rhsv=C^T y*(0) + mtxfv,
                                      umat=mtxtb, vmat=B~^T)[:NV]

    and the static ``feedbackthroughdict = {None: {...}}`` the closed-loop simulation reads
    (``static_feedback=True``).  Returns ``(feedbackthroughdict, Z)``."""
    pru = _pru if pru is None else pru
    lau = _lau if lau is None else lau
    NV = mmat.shape[0]
    nmat = sps.csr_matrix((NV, NV)) if convc_mat is None else convc_mat
    Z = pru.proj_alg_ric_newtonadi(mmat=mmat, amat=(-amat - nmat).tocsr(), jmat=jmat, bmat=tb_mat,
                                   wmat=trct_mat, z0=z0, nwtn_adi_dict=nwtn_adi_dict)["zfac"]
    if comprz_thresh is not None or comprz_maxc is not None:
        Z = pru.compress_Zsvd(Z, thresh=comprz_thresh, k=comprz_maxc)
    MT = mmat.T.tocsr()
    fv = np.zeros((NV, 1)) if fv is None else np.asarray(fv, dtype=float).reshape(NV, 1)
    mtxtb = -pru.get_mTzzTtb(MT, Z, tb_mat)
    mtxfv = -pru.get_mTzzTtb(MT, Z, fv)
    fl = mc_mat.T @ np.asarray(ystar0, dtype=float).reshape(-1, 1)
    w = lau.solve_sadpnt_smw(amat=(amat.T + nmat.T).tocsr(), jmat=jmat, rhsv=fl + mtxfv,
                             umat=mtxtb, vmat=sps.csr_matrix(tb_mat).T)[:NV]
    store.save(datastr + "__w", w)
    store.save(datastr + "__mtxtb", mtxtb)
    return {None: dict(w=datastr + "__w", mtxtb=datastr + "__mtxtb")}, Z


def simulate_linearized_flow(mmat=None, amat=None, jmat=None, tb_mat=None, rhsv=None,
                             iniv=None, tmesh=None, get_tdpart=None,
                             feedbackthroughdict=None, store=None, closed_loop=True,
                             static_feedback=False, lau=None):
    """Implicit Euler on ``tmesh``; returns ``{t: v(t)}`` (NV x 1 arrays).

    ``feedbackthroughdict`` maps ``t`` (or ``None`` for a static gain,
    ``optcont_main.py:516-521``) to the names of the stored ``w`` and ``mtxtb``;
    ``store`` resolves the names (see :mod:`optconpy_amd.dae_ric`).
    """
    lau = _lau if lau is None else lau
    NV = mmat.shape[0]
    tb = sps.csr_matrix(tb_mat) if tb_mat is not None else None
    v = np.zeros((NV, 1)) if iniv is None else np.asarray(iniv, dtype=float).reshape(NV, 1)
    rhsv = np.zeros((NV, 1)) if rhsv is None else np.asarray(rhsv, dtype=float).reshape(NV, 1)
    vels = {tmesh[0]: v.copy()}
    for k in range(len(tmesh) - 1):
        t1 = tmesh[k + 1]
        tau = t1 - tmesh[k]
        nmat, rhs_td = get_tdpart(time=t1) if get_tdpart is not None else (None, 0.0)
        op = (mmat / tau + amat) if nmat is None else (mmat / tau + amat + nmat)
        rhs = (mmat @ v) / tau + rhsv + rhs_td
        if closed_loop and feedbackthroughdict is not None:
            key = None if (static_feedback or t1 not in feedbackthroughdict) else t1
            gain = store.load(feedbackthroughdict[key]["mtxtb"])          # NV x NU
            w = store.load(feedbackthroughdict[key]["w"])
            rhs = rhs + tb @ (tb.T @ w)
            x = lau.solve_sadpnt_smw(amat=sps.csr_matrix(op), jmat=jmat, rhsv=rhs,
                                     umat=tb.toarray(), vmat=gain.T)
        else:
            x = lau.solve_sadpnt_smw(amat=sps.csr_matrix(op), jmat=jmat, rhsv=rhs)
        v = x[:NV]
        vels[t1] = v.copy()
    return vels


def eval_costfunc(V=None, W=None, cmat=None, ystar=None, tbmat=None, tmesh=None,
                  veldict=None, fbftdict=None, store=None, penau=True,
                  static_feedback=False):
    """``dy(T)' V dy(T) + int dy' W dy + u'u`` by the piecewise trapezoidal rule
    (``optcont_main.py:213-264``), ``dy = y* - C v``, ``u = mtxtb^T v + B~^T w``."""
    def dywdy(t, Wt):
        dy = ystar(t) - cmat @ veldict[t]
        return float(np.asarray(dy.T @ (Wt @ dy)).ravel()[0])

    def uru(t):
        if not penau or fbftdict is None:
            return 0.0
        # a static gain is stored under the key None (optcont_main.py:244-248: KeyError fallback)
        key = None if (static_feedback or t not in fbftdict) else t
        fb = store.load(fbftdict[key]["mtxtb"]).T @ veldict[t]
        ft = tbmat.T @ store.load(fbftdict[key]["w"])
        return float(np.asarray((fb + ft).T @ (fb + ft)).ravel()[0])

    cfv = 0.0
    old = dywdy(tmesh[0], W) + uru(tmesh[0])
    for k, t in enumerate(tmesh[1:]):
        new = dywdy(t, W) + uru(t)
        cfv += 0.5 * (t - tmesh[k]) * (new + old)
        old = new
    return cfv + dywdy(tmesh[-1], V)
