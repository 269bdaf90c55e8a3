"""GPU tests at the BASELINE.json configuration sizes (cfg2 ... cfg5) and the independent
dense pin of the HIP path.

* cfg2 (N = 58, n = 29 930, 16 shifts): the Newton-ADI solve the benchmark times, through the
  drop-in boundary, feedback gain K against the oracle's committed result
  (tests/golden/cfg2_golden.npz, made by tests/golden/make_golden.py --cfg2).
* cfg3 / cfg4 / cfg5 (n ~ 5e4 / 1e5 / 5e5): sizes where the oracle's 16 sparse LUs no longer
  fit a test budget -- size-independent properties instead: the TRUE residual of every column
  recomputed with scipy on the host (<= 1e-10 relative), the constraint J V = 0, linearity.
* small N: the HIP Newton-ADI against scipy.linalg.solve_continuous_are on ker(J)
  (tests/identities.py) for the three call forms of the reference -- steady
  (optcont_main.py:488-492), time step with transposed=True and z0, and with mtxoldb
  (solve_dae_ric.py:152-159).
"""
import os

import numpy as np
import pytest
import scipy.sparse as sps

from optconpy_amd import _lib, backend, problems as pb
from oracle import lin_alg_utils as olau, proj_ric_utils as opru

pytestmark = pytest.mark.gpu
K_TOL = 1e-6          # north_star parity bar
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


# ----------------------------------------------------------------------------- cfg2
def test_cfg2_newton_adi_gain_vs_oracle_fixture():
    """The benchmark's workload through the reference's own call
    (optcont_main.py:488-492 -> pru.proj_alg_ric_newtonadi, sweeps of 16 shifts): same number
    of Newton steps as the oracle, same update norms, K within the parity bar."""
    import sadptprj_riclyap_adi.lin_alg_utils as lau
    import sadptprj_riclyap_adi.proj_ric_utils as pru
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import CFG2
    g = np.load(os.path.join(ROOT, "tests", "golden", "cfg2_golden.npz"))
    backend.reset()
    pr = pb.ricc_problem(CFG2["N"], CFG2["nu"], NU=CFG2["NU"], NY=CFG2["NY"], alphau=CFG2["alphau"])
    chk = np.array([pr.M.data.sum(), pr.A.data.sum(), abs(pr.J.data).sum(), abs(pr.Nc.data).sum(),
                    pr.M.nnz, pr.A.nnz, pr.J.nnz, pr.Nc.nnz])
    assert np.allclose(g["mat_checks"], chk, rtol=1e-12)           # identical FEM matrices
    # operator preparation as optcont_main.py:405-425, on the GPU
    mct = lau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=pr.mc_mat.T, transposedprj=True)
    tb = lau.apply_invsqrt_fromright(pr.rmat, pr.b_mat, output="sparse")
    trct = lau.apply_invsqrt_fromright(pr.y_masmat, mct, output="dense")
    assert np.isclose(np.linalg.norm(trct), g["trct_fro"][0], rtol=1e-8)
    ms = pb.logshifts(CFG2["pmin"], CFG2["pmax"], CFG2["nshifts"])
    assert np.allclose(ms, g["shifts"])
    d = dict(pb.default_nwtn_adi_dict(), ms=ms)
    F = (-pr.A - pr.Nc).tocsr()
    out = pru.proj_alg_ric_newtonadi(mmat=pr.M, amat=F, jmat=pr.J, bmat=tb, wmat=trct, nwtn_adi_dict=d)
    assert out["gmres_nonconverged"] == 0
    assert out["nwtn_steps"] == int(g["nwtn_steps"][0])
    # the inner ADI stops after the same step as the oracle's step-by-step iteration in every
    # Newton step (the blocks of a sweep are the sequential blocks; tests/golden: 177 each)
    assert out["adi_steps"] == int(g["upd_hist"][:, 2].sum())
    # the last update norm is the difference of two nearly equal iterates: same decade as the
    # oracle's (the sweep form stops the inner ADI at sweep granularity)
    assert 0.1 * g["upd_hist"][-1, 1] <= out["upd_rel"] <= 10.0 * g["upd_hist"][-1, 1]
    K = -pru.get_mTzzTtb(pr.M.T, out["zfac"], tb)
    assert rel(K, g["K_ric"]) < K_TOL
    # ... and what bench.py times: ONE Newton step from the converged, compressed iterate
    Zc = pru.compress_Zsvd(out["zfac"], thresh=1e-9)
    one = pru.proj_alg_ric_newtonadi(mmat=pr.M, amat=F, jmat=pr.J, bmat=tb, wmat=trct, z0=Zc,
                                     nwtn_adi_dict=dict(d, nwtn_max_steps=1))
    assert rel(-pru.get_mTzzTtb(pr.M.T, one["zfac"], tb), g["K_ric"]) < K_TOL
    # first Newton step = open-loop Lyapunov solve (step-by-step ADI of the reference)
    lo = pru.solve_proj_lyap_stein(amat=F, mmat=pr.M, jmat=pr.J, wmat=trct, adi_dict=dict(d, sweep_width=16))
    assert rel(-pru.get_mTzzTtb(pr.M.T, lo["zfac"], tb), g["K_lyap"]) < K_TOL
    backend.reset()


# --------------------------------------------------------------------- cfg3 / cfg4 / cfg5
def _batched_properties(calA, calE, J, shifts, R, tol=1e-10, coarse_max=None, max_levels=None, info=None):
    """One batched solve of len(shifts) shifts against the shared panel R through the C-ABI;
    every property is recomputed on the host with scipy."""
    import torch
    nv, m = R.shape
    npr = J.shape[0]
    opts = {} if coarse_max is None else dict(coarse_max=coarse_max)
    if max_levels is not None:
        opts["max_levels"] = max_levels
    ctx = _lib.Context(0, **opts)
    ctx.set_operator(calA, calE, J)
    if info is not None:
        info.update(ctx.setup_info())
    dev = torch.device("cuda", 0)
    Rd = torch.as_tensor(R).to(dev)
    G = len(shifts)
    X = torch.empty(G, ctx.n, m, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    its, rr = ctx.shift_solve_batch_dev(shifts, [1.0] * G, Rd.data_ptr(), 0, m, X.data_ptr())
    ctx.synchronize()
    assert rr.max() <= tol * 1.0000001 and min(its) > 0
    Xh = X.cpu().numpy()
    del X
    bn = np.linalg.norm(R, axis=0)
    JT = J.T.tocsr()
    worst = 0.0
    for g, p in enumerate(shifts):
        V, L = Xh[g, :nv], Xh[g, nv:]
        rv = (calA @ V) + p * (calE @ V) + JT @ L - R          # velocity rows of S x - b
        rp = J @ V                                              # constraint rows
        res = np.sqrt(np.linalg.norm(rv, axis=0) ** 2 + np.linalg.norm(rp, axis=0) ** 2) / bn
        worst = max(worst, res.max())
        assert np.abs(rp).max() <= 1e-9 * np.abs(V).max()       # J V = 0
    assert worst <= 1.05 * tol, worst
    # linearity on one shift: S^-1 (2 r_0 - r_1) = 2 x_0 - x_1
    p = shifts[G // 2]
    comb = np.ascontiguousarray(2.0 * R[:, :1] - R[:, 1:2])
    Xc, _, _ = ctx.shift_solve(p, 1.0, comb)
    ref = 2.0 * Xh[G // 2][:, :1] - Xh[G // 2][:, 1:2]
    assert rel(Xc[:nv], ref[:nv]) < 1e-7
    ctx.close()
    return its, worst


def test_cfg3_batched_shift_solves_properties():
    """cfg3 surrogate: N = 75 (n = 50 177), nu = 0.15/40, 32 log-spaced shifts of which one
    sweep of 16 (every second) is solved in one batch, m = 16."""
    pr = pb.ricc_problem(75, 0.15 / 40.0)
    calA = (-pr.A - pr.Nc).T.tocsr()
    ms = pb.logshifts(1.0, 3e3, 32)[::2]
    R = np.random.default_rng(3).standard_normal((pr.NV, 16))
    its, worst = _batched_properties(calA, pr.M.T.tocsr(), pr.J, ms, R)
    print("cfg3: iterations per shift", its, "worst true residual %.2e" % worst)


def test_cfg4_dre_operator_wide_panel_properties():
    """cfg4 surrogate: N = 106 (n = 100 490), nu = 0.15/60, the time-varying DRE operator of
    solve_dae_ric.py:147  cal A = -(M^T/2 + tau (A^T + N^T))  with the largest step of the
    sine-squeezed mesh (optcont_main.py:141-150, Nts = 16), panel width m = 66
    (comprz_maxc + NY', solve_dae_ric.py:149), 16 of the 64 shifts in one batch."""
    pr = pb.ricc_problem(106, 0.15 / 60.0)
    tmesh = pb.get_tint(0.0, 1.0, 16, True)
    tau = float(np.diff(tmesh).max())
    MT = pr.M.T.tocsr()
    ft = (-(0.5 * MT + tau * (pr.A.T + pr.Nc.T))).tocsr()
    ms = pb.logshifts(0.5, 2e3, 64)[::4]
    R = np.random.default_rng(4).standard_normal((pr.NV, 66))
    info = {}
    its, worst = _batched_properties(ft, MT, pr.J, ms, R, info=info)
    assert info["levels"] == 3 and info["dense_coarse"] <= 4096 + 512     # the gentle third level serves this size
    print("cfg4: tau %.4f iterations per shift" % tau, its, "worst true residual %.2e" % worst)


def test_third_level_forced_small_problem(monkeypatch):
    """The child-level path (coarse problem handed to a second context instead of a dense inverse) on a
    problem small enough for the quick suite: N = 30 with coarse_max = 300 takes three levels, and the
    same solve with max_levels = 2 (aggregates grown instead) must not need fewer iterations -- like for like:
    plain aggregation on both (the two-level setup would otherwise smooth its aggregates, round 3)."""
    monkeypatch.setenv("RICADI_SA", "0")
    pr = pb.ricc_problem(30, 0.05)
    calA = (-pr.A - pr.Nc).T.tocsr()
    ms = [float(p) for p in pb.logshifts(1.0, 1e3, 8)]
    R = np.random.default_rng(6).standard_normal((pr.NV, 16))
    i3, i2 = {}, {}
    its3, w3 = _batched_properties(calA, pr.M.T.tocsr(), pr.J, ms, R, coarse_max=300, info=i3)
    its2, w2 = _batched_properties(calA, pr.M.T.tocsr(), pr.J, ms, R, coarse_max=300, max_levels=2, info=i2)
    assert i3["levels"] == 3 and i2["levels"] == 2, (i3, i2)
    assert i3["kc"] > 300 >= i2["kc"]
    assert sum(its3) <= sum(its2), (its3, its2)
    print("forced third level: iterations", its3, "two-level", its2)


def test_stiff_operator_keeps_two_levels_with_grown_aggregates(monkeypatch):
    """The hierarchy rule of round 4 (ricadi_host.cpp:build_setup, host side tested in test_capi_cpu.py): the
    diffusion-dominated operator whose coarse matrix exceeds coarse_max keeps TWO levels -- smoothed prolongation,
    dense inverse of up to 1.5 x coarse_max, aggregates grown for it -- and the batched solves meet the tolerance in
    the true residual; the same operator with plain aggregation (RICADI_SA=0) takes the child level and must not
    need fewer iterations (measured at n = 2e5: 2 142 vs 1 422 over 16 shifts)."""
    pr = pb.ricc_problem(30, 0.05)
    calA = (-pr.A - pr.Nc).T.tocsr()
    ms = [float(p) for p in pb.logshifts(1.0, 1e3, 8)]
    R = np.random.default_rng(6).standard_normal((pr.NV, 16))
    i2 = {}
    its2, w2 = _batched_properties(calA, pr.M.T.tocsr(), pr.J, ms, R, coarse_max=300, info=i2)
    assert i2["levels"] == 2 and 300 < i2["kc"] <= 450, i2
    plan = _lib.host_plan_levels(calA, pr.M.T.tocsr(), pr.J, coarse_max=300)
    assert plan["levels"] == 2 and plan["smoothed"] and plan["kc"] == i2["kc"], (plan, i2)
    monkeypatch.setenv("RICADI_SA", "0")
    i3 = {}
    its3, w3 = _batched_properties(calA, pr.M.T.tocsr(), pr.J, ms, R, coarse_max=300, info=i3)
    assert i3["levels"] == 3, i3
    assert sum(its2) <= sum(its3), (its2, its3)
    print("stiff operator: two levels (smoothed)", its2, "child level (plain)", its3)


def test_cfg5_batched_shift_solves_properties():
    """cfg5: N = 236 (NV = 443 682, NP = 56 168, n = 499 850, nnz(S) = 14.4e6), 16 of the
    128 shifts in one batch, m = 16."""
    pr = pb.ricc_problem(236, 0.05)
    assert pr.NV + pr.NP == 499850
    calA = (-pr.A - pr.Nc).T.tocsr()
    ms = pb.logshifts(1.0, 3e3, 128)[::8]
    R = np.random.default_rng(5).standard_normal((pr.NV, 16))
    its, worst = _batched_properties(calA, pr.M.T.tocsr(), pr.J, ms, R)
    print("cfg5: iterations per shift", its, "worst true residual %.2e" % worst)


# ------------------------------------------------------------ independent dense pin (GPU)
@pytest.mark.parametrize("N", [4, 8])
def test_hip_newton_adi_vs_dense_are_three_call_forms(N):
    """N = 4 (NV = 98) and N = 8 (NV = 450, 370 divergence-free directions): the HIP path against scipy's dense
    Riccati solver for the three call forms of the reference."""
    import sadptprj_riclyap_adi.proj_ric_utils as pru
    from identities import dense_projected_are, dre_step_inputs
    backend.reset()
    tight = dict(adi_max_steps=300, adi_newZ_reltol=1e-12, nwtn_max_steps=30, nwtn_upd_reltol=1e-11,
                 nwtn_upd_abstol=1e-14)
    # steady call
    pr = pb.ricc_problem(N, 0.2, NU=2, NY=2, alphau=1e-3)
    mct = olau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=pr.mc_mat.T, transposedprj=True)
    tb = olau.apply_invsqrt_fromright(pr.rmat, pr.b_mat, output="dense")
    trct = olau.apply_invsqrt_fromright(pr.y_masmat, mct, output="dense")
    F = (-pr.A - pr.Nc).tocsr()
    X = dense_projected_are(F.T, pr.M.T, pr.J, tb, trct)
    out = pru.proj_alg_ric_newtonadi(mmat=pr.M, amat=F, jmat=pr.J, bmat=tb, wmat=trct,
                                     nwtn_adi_dict=dict(tight, ms=pb.logshifts(1.0, 2e3, 10)))
    Z = out["zfac"]
    assert rel(Z @ Z.T, X) < 1e-6
    assert rel(pru.get_mTzzTtb(pr.M.T, Z, tb), pr.M.T @ (X @ tb)) < K_TOL
    # time-step call: transposed=True, z0, then also mtxoldb
    pr = pb.ricc_problem(N, 0.2, NU=2, NY=2, alphau=1e-2)
    for with_old in (False, True):
        kw, p = dre_step_inputs(pr, tau=0.05, with_old=with_old)
        B = np.sqrt(p["tau"]) * p["tb"]
        calA = p["ft"].toarray()
        if with_old:
            calA = calA + kw["mtxoldb"] @ B.T
        X = dense_projected_are(calA, p["MT"], pr.J, B, p["wmat"])
        out = pru.proj_alg_ric_newtonadi(nwtn_adi_dict=dict(tight, ms=pb.logshifts(0.4, 60.0, 8)), **kw)
        Z = out["zfac"]
        assert rel(Z @ Z.T, X) < 1e-6, with_old
        assert rel(p["MT"] @ (Z @ (Z.T @ B)), p["MT"] @ (X @ B)) < K_TOL, with_old
    backend.reset()
