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

__all__ = ["simulate_linearized_flow", "eval_costfunc"]


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
            key = None if static_feedback else t1
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
        key = None if static_feedback else t
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
