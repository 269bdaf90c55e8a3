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


def _pin_setup():
    """Two backward time steps at N = 8 (NV = 450), tight tolerances, no compression to speak of."""
    pr, kw, tmesh = _setup(N=8, Nts=2)
    kw = dict(kw, comprz_thresh=1e-10, comprz_maxc=400,
              nwtn_adi_dict=dict(kw["nwtn_adi_dict"], adi_max_steps=300, adi_newZ_reltol=1e-12, nwtn_max_steps=30,
                                 nwtn_upd_reltol=1e-11, nwtn_upd_abstol=1e-14, ms=pb.logshifts(0.4, 200.0, 10)))
    return pr, kw, tmesh


def test_two_step_sweep_oracle_vs_dense_riccati_solver():
    """The oracle's sweep pinned by dense mathematics: the gains of two backward time
    steps against scipy.linalg.solve_continuous_are on ker J, step by step (identities.dense_dre_sweep_gains) --
    no ADI, no Newton-Kleinman, no saddle-point solve shared."""
    from identities import dense_dre_sweep_gains
    pr, kw, tmesh = _pin_setup()
    dense = dense_dre_sweep_gains(pr, kw, tmesh)
    store = MemoryStore()
    fb = solve_flow_daeric(store=store, pru=opru, lau=olau, **kw)
    for t in tmesh:
        K = store.load(fb[t]["mtxtb"])
        assert np.linalg.norm(K - dense[t]) <= 1e-6 * np.linalg.norm(dense[t]), t


@pytest.mark.gpu
def test_two_step_sweep_gpu_vs_dense_riccati_solver():
    """The same pin for the MI355X modules: the product's sweep against the dense solver directly (not via the
    oracle), gain by gain at the 1e-6 bar."""
    from identities import dense_dre_sweep_gains
    from optconpy_amd import backend
    backend.reset()
    pr, kw, tmesh = _pin_setup()
    dense = dense_dre_sweep_gains(pr, kw, tmesh)
    store = MemoryStore()
    fb = solve_flow_daeric(store=store, **kw)
    for t in tmesh:
        K = store.load(fb[t]["mtxtb"])
        assert np.linalg.norm(K - dense[t]) <= 1e-6 * np.linalg.norm(dense[t]), t
    backend.reset()


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


@pytest.mark.gpu
def test_sweep_n30_gpu_vs_oracle_fixture():
    """The backward sweep of the differential Riccati equation at a size where the operator changes
    matter (N = 30, n = 7 922, four time steps on the sine-squeezed mesh, compression to 20 columns,
    z0 / w_mat hand-over between the steps): per time step the gain mtxtb and the feed-forward w against
    the ORACLE's sweep (tests/golden/dre30_golden.npz, made by make_golden.py --dre30: 4 minutes of
    oracle time, hence a fixture).  solve_dae_ric.py:121-211."""
    import os
    from optconpy_amd import backend
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dre30_golden.npz"))
    backend.reset()
    pr, kw, tmesh = _setup(N=30, Nts=4)
    assert np.allclose(tmesh, g["tmesh"])
    assert np.allclose(g["mat_checks"], [pr.M.data.sum(), pr.A.data.sum(), abs(pr.J.data).sum(),
                                         abs(pr.Nc.data).sum()], rtol=1e-12)
    sg = MemoryStore()
    fg = solve_flow_daeric(store=sg, **kw)             # MI355X modules
    for k, t in enumerate(tmesh):
        Kg, wg = sg.load(fg[t]["mtxtb"]), sg.load(fg[t]["w"])
        Ko, wo = g["mtxtb_%d" % k], g["w_%d" % k]
        assert np.linalg.norm(Kg - Ko) <= 1e-6 * np.linalg.norm(Ko), (k, t)
        assert np.linalg.norm(wg - wo) <= 1e-6 * np.linalg.norm(wo), (k, t)
    backend.reset()


# ------------------------------------------- outer-Newton accumulation (curnwtnsdict)
def _cns_dict(tmesh, prefix="cns"):
    """Names as optcont_main.py:201-210 (init_nwtnstps_value_dict) builds them."""
    return {t: dict(v="{0}__cns_v_t{1}".format(prefix, t), mtxtb="{0}__cns_mtxtb_t{1}".format(prefix, t),
                    w="{0}__cns_w_t{1}".format(prefix, t)) for t in tmesh}


class _SpyOld:
    """Records the mtxoldb / bmat arguments of every Newton-ADI call."""

    def __init__(self, mod):
        self.mod, self.seen = mod, []

    def __getattr__(self, name):
        f = getattr(self.mod, name)
        if name == "proj_alg_ric_newtonadi":
            def g(*a, **k):
                self.seen.append(None if k.get("mtxoldb") is None else np.array(k["mtxoldb"]))
                return f(*a, **k)
            return g
        return f


def _two_outer_steps(pru, lau, store):
    """Two passes of the sweep as the outer Newton loop of optcont_main.py:577-600 runs them:
    same curnwtnsdict, a new data string per pass (so that no Z is memoised)."""
    pr, kw, tmesh = _setup(N=5, Nts=3)
    cns = _cns_dict(tmesh)
    fbs = []
    for cnsno in range(2):
        kw2 = dict(kw, get_datastr=lambda time=None, **k: "cns{0}_t{1:.6f}".format(cnsno, time))
        fbs.append(solve_flow_daeric(store=store, pru=pru, lau=lau, curnwtnsdict=cns, **kw2))
    return pr, kw, tmesh, cns, fbs


def test_curnwtnsdict_accumulation_host_logic():
    """solve_dae_ric.py:133-141,151,181,197-200 with the oracle's modules: the first pass finds no
    stored feedback (IOError -> None) and stores gain(t_{k+1}) + gain(t_k) and w(t_k); the second
    pass hands sqrt(tau) * stored gain to the Newton-ADI call as mtxoldb and adds to the stored sums."""
    store = MemoryStore()
    spy = _SpyOld(opru)
    pr, kw, tmesh, cns, fbs = _two_outer_steps(spy, olau, store)
    nsteps = len(tmesh) - 1
    assert len(spy.seen) == 2 * nsteps
    assert all(s is None for s in spy.seen[:nsteps])               # pass 1: nothing stored yet
    assert all(s is not None for s in spy.seen[nsteps:])           # pass 2: feedback of pass 1
    # what pass 1 stored, from its own per-step results
    K1 = {t: store.load(fbs[0][t]["mtxtb"]) for t in tmesh}
    K2 = {t: store.load(fbs[1][t]["mtxtb"]) for t in tmesh}
    w1 = {t: store.load(fbs[0][t]["w"]) for t in tmesh}
    w2 = {t: store.load(fbs[1][t]["w"]) for t in tmesh}
    for k in range(nsteps):
        t, tn = tmesh[k], tmesh[k + 1]
        after1 = K1[tn] + K1[t]                                    # :181 then :199
        tau = tn - t
        # the order of the backward loop: spy.seen[nsteps + (nsteps-1-k)] belongs to time t
        got = spy.seen[nsteps + (nsteps - 1 - k)]
        assert np.allclose(got, np.sqrt(tau) * after1, rtol=1e-12, atol=1e-14)
        assert np.allclose(store.load(cns[t]["mtxtb"]), after1 + K2[tn] + K2[t], rtol=1e-10, atol=1e-14)
        assert np.allclose(store.load(cns[t]["w"]), w1[t] + w2[t], rtol=1e-10, atol=1e-14)
    # terminal entries are overwritten by every pass with the terminal values (:118-119)
    assert np.allclose(store.load(cns[tmesh[-1]]["mtxtb"]), K2[tmesh[-1]])
    # the old feedback changes the Riccati solution: pass 2 is not a copy of pass 1
    assert np.linalg.norm(K2[tmesh[0]] - K1[tmesh[0]]) > 1e-6 * np.linalg.norm(K1[tmesh[0]])


@pytest.mark.gpu
def test_curnwtnsdict_gpu_vs_oracle_modules():
    from optconpy_amd import backend, lin_alg_utils as glau, proj_ric_utils as gpru
    backend.reset()
    so, sg = MemoryStore(), MemoryStore()
    _, _, tmesh, cns, fo = _two_outer_steps(opru, olau, so)
    _, _, _, _, fg = _two_outer_steps(gpru, glau, sg)
    for cnsno in range(2):
        for t in tmesh:
            Ko, Kg = so.load(fo[cnsno][t]["mtxtb"]), sg.load(fg[cnsno][t]["mtxtb"])
            wo, wg = so.load(fo[cnsno][t]["w"]), sg.load(fg[cnsno][t]["w"])
            assert np.linalg.norm(Kg - Ko) <= 1e-6 * np.linalg.norm(Ko), (cnsno, t)
            assert np.linalg.norm(wg - wo) <= 1e-6 * np.linalg.norm(wo), (cnsno, t)
    for t in tmesh:
        assert np.linalg.norm(sg.load(cns[t]["mtxtb"]) - so.load(cns[t]["mtxtb"])) <= \
            1e-6 * np.linalg.norm(so.load(cns[t]["mtxtb"]))
    backend.reset()
