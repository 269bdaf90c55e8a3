"""Backward-in-time sweep of the differential-algebraic Riccati equation.

Python 3 counterpart of ``solve_flow_daeric`` (``/root/reference/solve_dae_ric.py:7-213``,
called from ``/root/reference/optcont_main.py:584-600``) on top of the MI355X
solver modules: per time step one Newton-ADI solve, one column compression,
the feedback gain and one feed-forward saddle-point solve -- all on the GPU
through :mod:`optconpy_amd.proj_ric_utils` / :mod:`optconpy_amd.lin_alg_utils`.
Same keyword arguments and the same returned ``feedbackthroughdict`` (time ->
names of the stored ``w`` and ``mtxtb``) as the reference; the per-step
results are memoised under the strings ``get_datastr`` produces, so an
interrupted sweep resumes where it stopped (``solve_dae_ric.py:143-170``).

Formulas, per step ``t_k -> t_{k+1}``, ``tau = t_{k+1} - t_k``
(``solve_dae_ric.py:147-194``):

    F_k    = -(M^T/2 + tau (A^T + N_k^T))                    coefficient of the ARE
    W_k    = [M^T Z_{k+1}, sqrt(tau) C~^T]                   right-hand-side factor
    Z_k    = compress(newton_adi(E = M^T, A = F_k, B = sqrt(tau) B~, z0 = Z_{k+1}))
    K_k    = -M^T Z_k Z_k^T B~                               ("mtxtb")
    w_k    = [[M^T + tau (A^T+N_k^T) - tau K_sum B~^T, J^T],[J, 0]]^-1
             (M^T w_{k+1} + tau (C~^T y*(t_k) - M^T Z_k Z_k^T f~))
"""
from __future__ import annotations

import os

import numpy as np

from . import lin_alg_utils as _lau
from . import proj_ric_utils as _pru

__all__ = ["solve_flow_daeric", "NpyStore", "MemoryStore"]


class NpyStore:
    """``.npy`` files named by the data strings (the reference's ``dou.save_npa`` /
    ``dou.load_npa``, ``solve_dae_ric.py:104-109``).  A missing entry raises ``IOError``."""

    def save(self, name, arr):
        np.save(name + ".npy", np.asarray(arr))

    def load(self, name):
        path = name + ".npy"
        if not os.path.exists(path):
            raise IOError("no stored array " + path)
        return np.load(path)

    def remove_matching(self, prefix, infix):
        """Delete the stored arrays ``prefix*infix*`` (``glob`` + ``os.remove``, optcont_main.py:207-208)."""
        import glob
        for f in glob.glob(glob.escape(prefix) + "*" + glob.escape(infix) + "*.npy"):
            os.remove(f)


class MemoryStore(dict):
    """In-memory stand-in for the file cache (tests, benchmarks)."""

    def save(self, name, arr):
        self[name] = np.array(arr, copy=True)

    def load(self, name):
        if name not in self:
            raise IOError("no stored array " + name)
        return self[name]

    def remove_matching(self, prefix, infix):
        for k in [k for k in self if k.startswith(prefix) and infix in k[len(prefix):]]:
            del self[k]


def solve_flow_daeric(mmat=None, amat=None, jmat=None, bmat=None,
                      cmat=None, rhsv=None, rhsp=None,
                      mcmat=None, v_is_my=False,
                      rmat=None, vmat=None,
                      gamma=1.0,
                      tmesh=None, ystarvec=None,
                      nwtn_adi_dict=None,
                      curnwtnsdict=None,
                      comprz_thresh=None, comprz_maxc=None, save_full_z=False,
                      get_tdpart=None, gttdprtargs=None,
                      get_datastr=None, gtdtstrargs=None,
                      check_c_consist=True,
                      store=None, pru=None, lau=None, verbose=False):
    """See the module docstring.  ``store`` defaults to :class:`NpyStore`;
    ``pru`` / ``lau`` default to the MI355X modules (the test-suite passes the
    CPU oracle's modules here to obtain reference values)."""
    pru = _pru if pru is None else pru
    lau = _lau if lau is None else lau
    store = NpyStore() if store is None else store
    gttdprtargs = {} if gttdprtargs is None else gttdprtargs
    gtdtstrargs = {} if gtdtstrargs is None else dict(gtdtstrargs)

    MT, AT = mmat.T.tocsr(), amat.T.tocsr()
    NV = amat.shape[0]

    # mcmat^T (or cmat^T) has to lie in the kernel of J M^-1 (solve_dae_ric.py:75-83)
    if check_c_consist:
        probe = mcmat if (v_is_my and mcmat is not None) else cmat
        if probe is not None:
            mic = lau.apply_massinv(MT, probe.T)
            if np.linalg.norm(jmat @ mic) > 1e-12 * max(1.0, np.linalg.norm(mic)) + 1e-12:
                raise Warning("the output matrix needs to be in the kernel of J*M.-1")

    def name(t):
        gtdtstrargs.update(time=t)
        return get_datastr(**gtdtstrargs)

    # weighted observation / control operators (solve_dae_ric.py:91-97)
    if v_is_my and mcmat is not None:
        tct = lau.apply_invsqrt_fromright(vmat, mcmat.T, output="dense")
    else:
        tct = lau.apply_sqrt_fromright(vmat, cmat.T, output="dense")
    tb = lau.apply_invsqrt_fromright(rmat, bmat, output="sparse")

    # terminal values (solve_dae_ric.py:100-119)
    tE = tmesh[-1]
    cur = name(tE)
    Zc = np.sqrt(gamma) * lau.apply_massinv(mmat, tct)
    mtxtb = -pru.get_mTzzTtb(MT, Zc, tb)
    store.save(cur + "__Z", Zc)
    store.save(cur + "__mtxtb", mtxtb)
    wc = None
    if ystarvec is not None:
        wc = lau.apply_massinv(MT, gamma * (mcmat.T @ ystarvec(tE)))
        store.save(cur + "__w", wc)
    feedbackthroughdict = {tE: dict(w=cur + "__w", mtxtb=cur + "__mtxtb")}
    if curnwtnsdict is not None:
        store.save(curnwtnsdict[tE]["w"], wc)
        store.save(curnwtnsdict[tE]["mtxtb"], mtxtb)

    for tk in range(len(tmesh) - 2, -1, -1):
        t = tmesh[tk]
        tau = tmesh[tk + 1] - t
        if verbose:
            print("Time is {0}, timestep is {1}".format(t, tau))
        cur = name(t)
        nmat, rhs_td = get_tdpart(time=t, **gttdprtargs)
        NT = nmat.T.tocsr()

        # feedback accumulated by earlier outer Newton steps (solve_dae_ric.py:133-141)
        old_w, old_gain = None, None
        if curnwtnsdict is not None:
            try:
                old_w = store.load(curnwtnsdict[t]["w"])
                old_gain = store.load(curnwtnsdict[t]["mtxtb"])
            except IOError:
                old_w, old_gain = None, None

        try:
            Zc = store.load(cur + "__Z")
        except IOError:
            ft = -(0.5 * MT + tau * (AT + NT))
            wfac = np.hstack([MT @ Zc, np.sqrt(tau) * tct])
            Zp = pru.proj_alg_ric_newtonadi(
                mmat=MT, amat=ft, transposed=True,
                mtxoldb=None if old_gain is None else np.sqrt(tau) * old_gain,
                jmat=jmat, bmat=np.sqrt(tau) * tb, wmat=wfac, z0=Zc,
                nwtn_adi_dict=nwtn_adi_dict)["zfac"]
            if comprz_maxc is not None or comprz_thresh is not None:
                Zc = pru.compress_Zsvd(Zp, thresh=comprz_thresh, k=comprz_maxc)
            else:
                Zc = Zp
            store.save(cur + "__Z", Zp if save_full_z else Zc)

        # affine part: feed-forward w (solve_dae_ric.py:172-194)
        at = MT + tau * (AT + NT)
        ftilde = rhs_td + rhsv
        if old_w is not None:
            ftilde = ftilde + old_w
        gain_sum = mtxtb if old_gain is None else old_gain + mtxtb
        mtxft = pru.get_mTzzTtb(MT, Zc, ftilde)
        rhs_w = MT @ wc + tau * (mcmat.T @ ystarvec(t) - mtxft)
        mtxtb = -pru.get_mTzzTtb(MT, Zc, tb)
        wc = lau.solve_sadpnt_smw(amat=at, jmat=jmat, umat=tau * gain_sum, vmat=tb.T,
                                  rhsv=rhs_w)[:NV]

        if curnwtnsdict is not None:
            # as in the reference (solve_dae_ric.py:181,197-200) the stored sum
            # carries both the gain of the following time instance and the new one
            new_w = wc if old_w is None else old_w + wc
            new_gain = gain_sum + mtxtb
            store.save(curnwtnsdict[t]["w"], new_w)
            store.save(curnwtnsdict[t]["mtxtb"], new_gain)
        store.save(cur + "__w", wc)
        store.save(cur + "__mtxtb", mtxtb)
        feedbackthroughdict[t] = dict(w=cur + "__w", mtxtb=cur + "__mtxtb")

    return feedbackthroughdict
