"""Closed-loop forward simulation with the computed feedback, and the cost functional.

Counterpart of what ``optcon_nse`` does after the Riccati solve
(``/root/reference/optcont_main.py:529-536,609-626``: ``snu.solve_nse(closed_loop=True,
feedbackthroughdict=..., tb_mat=...)``) and of ``eval_costfunc``
(``optcont_main.py:213-264``).  ``snu`` (dolfin_navier_scipy) is not available,
so the time steppers here are this repo's own implicit Euler schemes: :func:`simulate_nse_flow` for the
nonlinear flow (Picard / Newton steps per time step over the discrete convection linearisation
:func:`problems.convection_from_vector`), and for the *linearised* flow (the reference's
``linearized_nse=True`` branch):

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

__all__ = ["simulate_linearized_flow", "simulate_nse_flow", "get_tdpart_from_velocities", "eval_costfunc",
           "steady_state_feedback"]


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


def get_tdpart_from_velocities(N, dictofvalues, store=None, ordering="component"):
    """The ``get_tdpart`` of the time-dependent branch (``optcont_main.py:556-568``): the convection
    linearisation about the velocity STORED for that time -- ``dictofvalues[time]`` is the array itself or
    its name in ``store`` (the reference keeps file names, ``dou.load_npa(dictofvalues[time])``).
    Returns ``get_tdpart(time=...) -> (convc_mat, rhs_con)`` as ``solve_flow_daeric`` and the flow
    simulations consume it (``solve_dae_ric.py:131``)."""
    from . import problems as pb

    def get_tdpart(time=None, **kw):
        cur = dictofvalues[time]
        if isinstance(cur, str):
            cur = store.load(cur)
        convc_mat, rhs_con = pb.convection_from_vector(N, cur, ordering=ordering)
        return convc_mat, rhs_con
    return get_tdpart


def simulate_nse_flow(mmat=None, amat=None, jmat=None, N=None, tb_mat=None, rhsv=None, iniv=None,
                      tmesh=None, feedbackthroughdict=None, store=None, closed_loop=False,
                      static_feedback=False, vel_pcrd_stps=1, vel_nwtn_stps=2, ordering="component",
                      lau=None, return_info=False, lin_vel_point=None):
    """Implicit Euler for the NONLINEAR Navier-Stokes flow, optionally in closed loop -- the
    ``snu.solve_nse(closed_loop=True, feedbackthroughdict=..., vel_pcrd_stps=1, vel_nwtn_stps=2)`` of
    ``optcont_main.py:609-626`` (and, without feedback, the forward solve of ``:548-550`` whose stored
    velocities feed ``get_tdpart``):

        M (v1 - v0)/tau + A v1 + H(v1) + J^T p = f + B~ u(t1),   J v1 = 0,
        u = mtxtb(t1)^T v1 + B~^T w(t1)                          (closed loop)

    Per time step ``vel_pcrd_stps`` Picard steps, ``H(v) ~ (vbar . grad) v``, then ``vel_nwtn_stps`` Newton
    steps, ``H(v) ~ N(vbar) v - H(vbar)``, each one saddle-point solve through ``lau.solve_sadpnt_smw``
    (the feedback as its low-rank term, the call of ``solve_dae_ric.py:192-194``), starting from the
    previous time step.  The convection operators come from :func:`problems.convection_from_vector`
    (the discrete ``snu.get_v_conv_conts``).  Returns ``{t: v(t)}`` (and, with ``return_info``, the
    norm of the last Newton update of every step)."""
    from . import problems as pb
    lau = _lau if lau is None else lau
    NV = mmat.shape[0]
    tb = sps.csr_matrix(tb_mat) if tb_mat is not None else None
    v = np.zeros((NV, 1)) if iniv is None else np.asarray(iniv, dtype=float).reshape(NV, 1)
    rhsv = np.zeros((NV, 1)) if rhsv is None else np.asarray(rhsv, dtype=float).reshape(NV, 1)
    vels = {tmesh[0]: v.copy()}
    last_upd = {}
    for k in range(len(tmesh) - 1):
        t1 = tmesh[k + 1]
        tau = t1 - tmesh[k]
        base = (mmat @ v) / tau + rhsv
        gain = w = None
        if closed_loop and feedbackthroughdict is not None:
            key = None if (static_feedback or t1 not in feedbackthroughdict) else t1
            gain = store.load(feedbackthroughdict[key]["mtxtb"])          # NV x NU
            w = store.load(feedbackthroughdict[key]["w"])
            base = base + tb @ (tb.T @ w)
        # linearisation point of the first step of this time step: the previous time step's velocity, or the
        # given trajectory (`lin_vel_point`: {t: v}, optcont_main.py:612 -- the linearised NSE about the last flow)
        vbar = v.copy() if lin_vel_point is None else np.asarray(lin_vel_point[t1], dtype=float).reshape(NV, 1)
        upd = 0.0
        for it in range(int(vel_pcrd_stps) + int(vel_nwtn_stps)):
            newton = it >= int(vel_pcrd_stps)
            nmat, hv = pb.convection_from_vector(N, vbar, ordering=ordering, newton_term=newton)
            op = sps.csr_matrix(mmat / tau + amat + nmat)
            rhs = base + hv if newton else base
            if gain is not None:
                x = lau.solve_sadpnt_smw(amat=op, jmat=jmat, rhsv=rhs, umat=tb.toarray(), vmat=gain.T)
            else:
                x = lau.solve_sadpnt_smw(amat=op, jmat=jmat, rhsv=rhs)
            upd = float(np.linalg.norm(x[:NV] - vbar))
            vbar = x[:NV]
        v = vbar
        vels[t1] = v.copy()
        last_upd[t1] = upd
    return (vels, dict(last_update=last_upd)) if return_info else vels


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
