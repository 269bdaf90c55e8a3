"""The outer Newton iteration of optcon_nse's time-dependent branch as PRODUCT code
(optconpy_amd/outer_newton.py; /root/reference/optcont_main.py:201-210,538-626).

CPU: host logic with the oracle's modules (accumulators, stale-file removal, the iteration contracts).
GPU: the same two outer steps through the MI355X modules against the oracle-module run.
"""
import numpy as np
import pytest

from optconpy_amd import problems as pb
from optconpy_amd.dae_ric import MemoryStore, NpyStore
from optconpy_amd.outer_newton import init_nwtnstps_value_dict, outer_newton_flow_control
from oracle import lin_alg_utils as olau, proj_ric_utils as opru


def _args(N=5, Nts=3):
    pr = pb.ricc_problem(N, 0.2, NU=2, NY=2, alphau=1e-2)
    mct = olau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=pr.mc_mat.T, transposedprj=True)
    tmesh = pb.get_tint(0.0, 0.3, Nts, True)
    nad = dict(pb.default_nwtn_adi_dict(), ms=pb.logshifts(0.6, 40.0, 6), adi_max_steps=120, nwtn_max_steps=6)
    NY2 = mct.shape[1]
    iniv = 0.5 * pb.nodal_interpolant(N)
    iniv = olau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=pr.M @ iniv, transposedprj=True)
    iniv = olau.apply_massinv(pr.M.T.tocsr(), iniv)
    kw = dict(mmat=pr.M, amat=pr.A, jmat=pr.J, N=N, bmat=pr.b_mat, mcmat=mct.T, rmat=pr.rmat, vmat=pr.y_masmat,
              gamma=1e-1, tmesh=tmesh, iniv=iniv, nwtn_adi_dict=nad, comprz_thresh=5e-5, comprz_maxc=20,
              ystarvec=lambda t: (0.1 * np.sin(5 * 3.14 * t) * np.arange(1, NY2 + 1)).reshape(-1, 1))
    return pr, kw, tmesh


def test_init_nwtnstps_value_dict_names_and_stale_entries(tmp_path):
    tmesh = [0.0, 0.5, 1.0]
    ms = MemoryStore()
    ms.save("run__cns_w_t0.5", np.ones(2))
    ms.save("run__Z_t0.5", np.ones(2))
    cnd = init_nwtnstps_value_dict(tmesh=tmesh, data_prfx="run", store=ms)
    assert cnd[0.5] == dict(v="run__cns_v_t0.5", mtxtb="run__cns_mtxtb_t0.5", w="run__cns_w_t0.5")
    assert "run__cns_w_t0.5" not in ms and "run__Z_t0.5" in ms        # only the accumulators go (:207-208)
    fs = NpyStore()
    pre = str(tmp_path / "run")
    fs.save(pre + "__cns_mtxtb_t1.0", np.ones(3))
    fs.save(pre + "__w_t1.0", np.ones(3))
    init_nwtnstps_value_dict(tmesh=tmesh, data_prfx=pre, store=fs)
    with pytest.raises(IOError):
        fs.load(pre + "__cns_mtxtb_t1.0")
    assert fs.load(pre + "__w_t1.0").shape == (3,)


def _run(pru, lau, linearized, steps=2):
    pr, kw, tmesh = _args()
    store = MemoryStore()
    fb, vels, hist = outer_newton_flow_control(outernwtnstps=steps, linearized_nse=linearized, store=store,
                                               pru=pru, lau=lau, data_prfx="t", **kw)
    return pr, tmesh, store, fb, vels, hist


@pytest.mark.parametrize("linearized", [False, True])
def test_outer_newton_host_logic(linearized):
    pr, tmesh, store, fb, vels, hist = _run(opru, olau, linearized, steps=3)
    assert sorted(fb) == sorted(tmesh.tolist()) and sorted(vels) == sorted(tmesh.tolist())
    assert all(np.abs(pr.J @ v).max() < 1e-10 for v in vels.values())            # divergence free
    # the accumulators hold the SUM over the outer steps (solve_dae_ric.py:181,197-200) ...
    acc = store.load("t__cns_mtxtb_t{0}".format(tmesh[0]))
    assert acc.shape == store.load(fb[tmesh[0]]["mtxtb"]).shape
    # ... every pass stored its own sweep under its own data string, nothing was memoised across passes
    assert all(any(k.startswith("t_cns{0}_t".format(c)) for k in store) for c in range(3))
    # the iteration contracts: the change of the flow trajectory falls from step to step
    chg = [h["velocity_change"] for h in hist]
    assert chg[1] < chg[0] and chg[2] < chg[1], chg


@pytest.mark.gpu
@pytest.mark.parametrize("linearized", [False, True])
def test_outer_newton_gpu_vs_oracle_modules(linearized):
    from optconpy_amd import backend, lin_alg_utils as glau, proj_ric_utils as gpru
    backend.reset()
    _, tmesh, so, fo, vo, ho = _run(opru, olau, linearized)
    _, _, sg, fg, vg, hg = _run(gpru, glau, linearized)
    for t in tmesh:
        Ko, Kg = so.load(fo[t]["mtxtb"]), sg.load(fg[t]["mtxtb"])
        assert np.linalg.norm(Kg - Ko) <= 1e-6 * np.linalg.norm(Ko), t
        assert np.linalg.norm(vg[t] - vo[t]) <= 1e-6 * max(np.linalg.norm(vo[t]), 1e-30), t
    assert abs(hg[-1]["velocity_change"] - ho[-1]["velocity_change"]) <= 1e-5 * ho[0]["velocity_change"]
    backend.reset()
