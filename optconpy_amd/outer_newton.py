"""The outer Newton iteration of the time-dependent branch of ``optcon_nse``.

Python 3 counterpart of ``/root/reference/optcont_main.py:538-626``: the nonlinear optimal-control problem is
solved by alternating

* a backward sweep of the differential Riccati equation linearised about the current flow
  (:func:`optconpy_amd.dae_ric.solve_flow_daeric`, ``optcont_main.py:584-600``), which ACCUMULATES the gains and
  feed-forward terms of the earlier passes in ``curnwtnsdict`` (``init_nwtnstps_value_dict``,
  ``optcont_main.py:201-210``; ``solve_dae_ric.py:133-141,181,197-200``), and
* a forward closed-loop simulation whose velocities become the next linearisation point
  (``optcont_main.py:609-626``: ``vel_nwtn_stps=1, vel_pcrd_stps=0`` about ``lin_vel_point`` for the linearised
  NSE, ``vel_pcrd_stps=1, vel_nwtn_stps=2`` for the nonlinear one),

``outernwtnstps`` times.  Everything that solves runs through the MI355X modules (or whatever ``pru`` / ``lau`` the
caller injects -- the test-suite passes the CPU oracle's to obtain reference values).  FEM assembly is replaced by
:mod:`optconpy_amd.problems` (the reference needs FEniCS).
"""
from __future__ import annotations

import numpy as np

from . import lin_alg_utils as _lau
from . import proj_ric_utils as _pru
from .closed_loop import get_tdpart_from_velocities, simulate_nse_flow
from .dae_ric import NpyStore, solve_flow_daeric

__all__ = ["init_nwtnstps_value_dict", "outer_newton_flow_control"]


def init_nwtnstps_value_dict(tmesh=None, data_prfx=None, store=None):
    """Names of the per-time accumulators of the outer Newton iteration, ``{t: {'v', 'mtxtb', 'w'}}``, and removal
    of whatever an earlier run left under them (``optcont_main.py:201-210``)."""
    cnd = {}
    for t in tmesh:
        cnd[t] = dict(v=data_prfx + "__cns_v_t{0}".format(t),
                      mtxtb=data_prfx + "__cns_mtxtb_t{0}".format(t),
                      w=data_prfx + "__cns_w_t{0}".format(t))
    if store is not None and hasattr(store, "remove_matching"):
        store.remove_matching(data_prfx, "__cns_")
    return cnd


def outer_newton_flow_control(mmat=None, amat=None, jmat=None, N=None, bmat=None, mcmat=None, rmat=None,
                              vmat=None, rhsv=None, gamma=1.0, tmesh=None, ystarvec=None, iniv=None,
                              nwtn_adi_dict=None, comprz_thresh=None, comprz_maxc=None, outernwtnstps=2,
                              linearized_nse=False, data_prfx="optcont", ordering="component",
                              store=None, pru=None, lau=None, verbose=False):
    """``optcont_main.py:538-626`` for the driven-cavity problems of :mod:`optconpy_amd.problems`.

    Returns ``(feedbackthroughdict, dictofvels, history)``: the feedback of the LAST backward sweep (time -> names
    of ``w`` and ``mtxtb`` in ``store``), the velocities of the last closed-loop simulation (time -> array), and per
    outer step the norm of the change of the velocity trajectory (what the iteration converges in)."""
    pru = _pru if pru is None else pru
    lau = _lau if lau is None else lau
    store = NpyStore() if store is None else store
    NV = mmat.shape[0]
    rhsv = np.zeros((NV, 1)) if rhsv is None else rhsv
    tb = lau.apply_invsqrt_fromright(rmat, bmat, output="sparse")
    sim = dict(mmat=mmat, amat=amat, jmat=jmat, N=N, tb_mat=tb, rhsv=rhsv, tmesh=tmesh, iniv=iniv,
               ordering=ordering, lau=lau)
    # forward solve without control: the first linearisation point (optcont_main.py:548-550)
    vels = simulate_nse_flow(closed_loop=False, vel_nwtn_stps=3, **sim)

    def store_vels(vd, tag):
        names = {}
        for t, v in vd.items():
            names[t] = "{0}__vel_{1}_t{2:.6f}".format(data_prfx, tag, t)
            store.save(names[t], v)
        return names

    curnwtnsdict = init_nwtnstps_value_dict(tmesh=tmesh, data_prfx=data_prfx, store=store)
    fb, history = None, []
    for cns in range(int(outernwtnstps)):
        names = store_vels(vels, "cns{0}".format(cns))
        prfx = "{0}_cns{1}".format(data_prfx, cns)
        fb = solve_flow_daeric(
            mmat=mmat, amat=amat, jmat=jmat, bmat=bmat, mcmat=mcmat, v_is_my=True, rmat=rmat, vmat=vmat,
            rhsv=rhsv, gamma=gamma, tmesh=tmesh, ystarvec=ystarvec, nwtn_adi_dict=nwtn_adi_dict,
            comprz_thresh=comprz_thresh, comprz_maxc=comprz_maxc, save_full_z=False,
            get_tdpart=get_tdpart_from_velocities(N, names, store=store, ordering=ordering),
            curnwtnsdict=curnwtnsdict,
            get_datastr=lambda time=None, **k: "{0}_t{1:.6f}".format(prfx, time), gtdtstrargs={},
            store=store, pru=pru, lau=lau, verbose=verbose)
        # closed-loop forward simulation = the next linearisation point (optcont_main.py:609-626)
        if linearized_nse:
            new = simulate_nse_flow(feedbackthroughdict=fb, store=store, closed_loop=True,
                                    vel_pcrd_stps=0, vel_nwtn_stps=1, lin_vel_point=vels, **sim)
        else:
            new = simulate_nse_flow(feedbackthroughdict=fb, store=store, closed_loop=True,
                                    vel_pcrd_stps=1, vel_nwtn_stps=2, **sim)
        chg = np.sqrt(sum(float(np.linalg.norm(new[t] - vels[t]) ** 2) for t in tmesh))
        history.append(dict(cns=cns, velocity_change=chg,
                            gain_norm_t0=float(np.linalg.norm(store.load(fb[tmesh[0]]["mtxtb"])))))
        if verbose:
            print("outer Newton step {0}: |v_new - v_old| = {1:.3e}".format(cns, chg))
        vels = new
    return fb, vels, history
