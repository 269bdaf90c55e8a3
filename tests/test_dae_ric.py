"""Counterpart of solve_flow_daeric (/root/reference/solve_dae_ric.py:7-213).

CPU part: host logic (terminal values, memoisation / resume) with the oracle's
modules injected.  GPU part: the same sweep through the MI355X modules against
the oracle-module sweep, per time step.
"""
import numpy as np
import pytest

from optconpy_amd import problems as pb
from optconpy_amd.dae_ric import MemoryStore, solve_flow_daeric
from oracle import lin_alg_utils as olau, proj_ric_utils as opru


def _setup(N=5, Nts=3):
    pr = pb.ricc_problem(N, 0.2, NU=2, NY=2, alphau=1e-2)
    mct = olau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=pr.mc_mat.T, transposedprj=True)
    tmesh = pb.get_tint(0.0, 0.3, Nts, True)
    nad = dict(pb.default_nwtn_adi_dict(), ms=pb.logshifts(0.6, 40.0, 6), adi_max_steps=120,
               nwtn_max_steps=6)
    NY2 = mct.shape[1]

    def ystar(t):
        return (0.1 * np.sin(5 * 3.14 * t) * np.arange(1, NY2 + 1)).reshape(-1, 1)

    def tdpart(time=None):
        return (1.0 + time) * pr.Nc, np.zeros((pr.NV, 1))

    def datastr(time=None, **kw):
        return "dre_t{0:.6f}".format(time)

    kw = dict(mmat=pr.M, amat=pr.A, jmat=pr.J, bmat=pr.b_mat, mcmat=mct.T, v_is_my=True,
              rmat=pr.rmat, vmat=pr.y_masmat, rhsv=np.zeros((pr.NV, 1)), gamma=1e-1,
              tmesh=tmesh, ystarvec=ystar, nwtn_adi_dict=nad, comprz_thresh=5e-5,
              comprz_maxc=20, get_tdpart=tdpart, get_datastr=datastr, gtdtstrargs={})
    return pr, kw, tmesh


class _Counting:
    def __init__(self, mod):
        self.mod, self.calls = mod, 0

    def __getattr__(self, name):
        f = getattr(self.mod, name)
        if name == "proj_alg_ric_newtonadi":
            def g(*a, **k):
                self.calls += 1
                return f(*a, **k)
            return g
        return f


def test_host_logic_terminal_values_and_resume():
    pr, kw, tmesh = _setup()
    store = MemoryStore()
    cnt = _Counting(opru)
    fb = solve_flow_daeric(store=store, pru=cnt, lau=olau, **kw)
    assert sorted(fb) == sorted(tmesh.tolist()) and cnt.calls == len(tmesh) - 1
    # terminal values (solve_dae_ric.py:100-108): Z(T) = sqrt(gamma) M^-1 C~^T, w(T) = M^-T gamma C^T y*(T)
    tct = olau.apply_invsqrt_fromright(kw["vmat"], kw["mcmat"].T, output="dense")
    ZT = store.load(fb[tmesh[-1]]["mtxtb"].replace("__mtxtb", "__Z"))
    assert np.allclose(pr.M @ ZT, np.sqrt(kw["gamma"]) * tct, atol=1e-12)
    wT = store.load(fb[tmesh[-1]]["w"])
    assert np.allclose(pr.M.T @ wT, kw["gamma"] * (kw["mcmat"].T @ kw["ystarvec"](tmesh[-1])), atol=1e-12)
    # gains have the shape of B~ and stay in the projected space: J M^-T K = 0
    K0 = store.load(fb[tmesh[0]]["mtxtb"])
    assert K0.shape == (pr.NV, kw["bmat"].shape[1])
    assert np.abs(pr.J @ olau.apply_massinv(pr.M, K0)).max() < 1e-9 * max(np.abs(K0).max(), 1e-30)
    # resume: every Z is memoised -> a second sweep does no Newton-ADI solve
    cnt2 = _Counting(opru)
    fb2 = solve_flow_daeric(store=store, pru=cnt2, lau=olau, **kw)
    assert cnt2.calls == 0
    assert np.allclose(store.load(fb2[tmesh[0]]["w"]), store.load(fb[tmesh[0]]["w"]))
    # a missing entry is recomputed (IOError path, solve_dae_ric.py:143-146)
    del store[fb[tmesh[1]]["mtxtb"].replace("__mtxtb", "__Z")]
    cnt3 = _Counting(opru)
    solve_flow_daeric(store=store, pru=cnt3, lau=olau, **kw)
    assert cnt3.calls == 1


def test_c_consistency_check_raises():
    pr, kw, tmesh = _setup(Nts=1)
    bad = dict(kw, mcmat=pr.mc_mat)        # not projected -> J M^-1 mcmat^T != 0
    with pytest.raises(Warning):
        solve_flow_daeric(store=MemoryStore(), pru=opru, lau=olau, **bad)


@pytest.mark.gpu
def test_sweep_gpu_vs_oracle():
    from optconpy_amd import backend
    backend.reset()
    pr, kw, tmesh = _setup()
    so, sg = MemoryStore(), MemoryStore()
    fo = solve_flow_daeric(store=so, pru=opru, lau=olau, **kw)
    fg = solve_flow_daeric(store=sg, **kw)             # MI355X modules
    for t in tmesh:
        Ko, Kg = so.load(fo[t]["mtxtb"]), sg.load(fg[t]["mtxtb"])
        wo, wg = so.load(fo[t]["w"]), sg.load(fg[t]["w"])
        assert np.linalg.norm(Kg - Ko) <= 1e-6 * np.linalg.norm(Ko), t
        assert np.linalg.norm(wg - wo) <= 1e-6 * np.linalg.norm(wo), t
    backend.reset()
