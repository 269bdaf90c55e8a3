"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle.

FP64 throughout.  Tolerances: kernels 1e-12 relative; solves at the GMRES
tolerance; the feedback gain K within 1e-6 relative Frobenius norm -- the bar
BASELINE.json's north_star states -- and in practice ~1e-11.
"""
import numpy as np
import pytest
import scipy.sparse as sps

from optconpy_amd import _lib, backend, problems as pb
from oracle import lin_alg_utils as olau, proj_ric_utils as opru

pytestmark = pytest.mark.gpu
K_TOL = 1e-6          # north_star parity bar


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def ctx1(cfg1):
    pr, tb, trct, ms = cfg1
    F = (-pr.A - pr.Nc).tocsr()
    ctx = _lib.Context(0)
    ctx.set_operator(F.T.tocsr(), pr.M.T.tocsr(), pr.J)
    yield ctx
    ctx.close()


# --------------------------------------------------------------------------- K1
@pytest.mark.parametrize("m", [1, 5, 16, 17, 48, 70, 128])
def test_spmm_parity(ctx1, cfg1, m):
    pr, tb, trct, ms = cfg1
    calA = (-pr.A - pr.Nc).T.tocsr()
    rng = np.random.default_rng(m)
    X = rng.standard_normal((pr.NV + pr.NP, m))
    for (al, be) in ((-3.0, 1.0), (1.0, 0.0), (0.0, 1.0)):
        S = sps.bmat([[be * calA + al * pr.M, pr.J.T], [pr.J, None]], format="csr")
        assert rel(ctx1.spmm(al, be, X), S @ X) < 1e-13


def test_spmm_lowrank_term(ctx1, cfg1):
    pr, tb, trct, ms = cfg1
    calA = (-pr.A - pr.Nc).T.tocsr()
    rng = np.random.default_rng(7)
    U = rng.standard_normal((pr.NV, 8))
    V = rng.standard_normal((pr.NV, 8))
    X = rng.standard_normal((pr.NV + pr.NP, 16))
    ctx1.set_lowrank(U, V)
    try:
        Y = ctx1.spmm(-2.0, 1.0, X)
    finally:
        ctx1.set_lowrank(None, None)
    S = sps.bmat([[calA - 2.0 * pr.M, pr.J.T], [pr.J, None]], format="csr")
    ref = S @ X
    ref[:pr.NV] -= U @ (V.T @ X[:pr.NV])
    assert rel(Y, ref) < 1e-12


def test_linearity_of_spmm(ctx1, cfg1):
    pr = cfg1[0]
    rng = np.random.default_rng(11)
    X1 = rng.standard_normal((pr.NV + pr.NP, 16))
    X2 = rng.standard_normal((pr.NV + pr.NP, 16))
    Y = ctx1.spmm(-5.0, 1.0, 2.0 * X1 - 3.0 * X2)
    assert rel(Y, 2.0 * ctx1.spmm(-5.0, 1.0, X1) - 3.0 * ctx1.spmm(-5.0, 1.0, X2)) < 1e-13


# ------------------------------------------------------------------ shift solves
@pytest.mark.parametrize("p", [-1.0, -30.0, -1000.0])
def test_shift_solve_vs_lu(ctx1, cfg1, p):
    pr = cfg1[0]
    calA = (-pr.A - pr.Nc).T.tocsr()
    rng = np.random.default_rng(3)
    R = rng.standard_normal((pr.NV, 16))
    X, its, rr = ctx1.shift_solve(p, 1.0, R)
    Xo = olau.SaddleLU(calA + p * pr.M, pr.J).solve(R)
    assert rr.max() < 1e-10 and its > 0
    assert rel(X[:pr.NV], Xo[:pr.NV]) < 1e-9
    assert np.abs(pr.J @ X[:pr.NV]).max() < 1e-9 * np.abs(X[:pr.NV]).max()   # J V = 0


def test_shift_solve_edge_panels(ctx1, cfg1):
    """single column, zero columns, nonzero constraint rhs."""
    pr = cfg1[0]
    calA = (-pr.A - pr.Nc).T.tocsr()
    rng = np.random.default_rng(4)
    lu = olau.SaddleLU(calA - 10.0 * pr.M, pr.J)
    r1 = rng.standard_normal((pr.NV, 1))
    X, _, _ = ctx1.shift_solve(-10.0, 1.0, r1)
    assert rel(X, lu.solve(r1)) < 1e-8
    R = rng.standard_normal((pr.NV, 4))
    R[:, 2] = 0.0                                   # a zero column must stay zero
    X, _, rr = ctx1.shift_solve(-10.0, 1.0, R)
    assert np.all(X[:, 2] == 0.0) and np.isfinite(X).all()
    Rp = rng.standard_normal((pr.NP, 3))
    Rv = rng.standard_normal((pr.NV, 3))
    X, _, _ = ctx1.shift_solve(-10.0, 1.0, Rv, Rp)
    assert rel(X, lu.solve(Rv, Rp)) < 1e-8


def test_shift_solve_wide_panel(ctx1, cfg1):
    """w_mat of the DRE sweep is up to comprz_maxc + NY' ~ 70 columns wide
    (solve_dae_ric.py:149); panels beyond 128 columns are split by the host layer."""
    pr = cfg1[0]
    calA = (-pr.A - pr.Nc).T.tocsr()
    rng = np.random.default_rng(8)
    lu = olau.SaddleLU(calA - 30.0 * pr.M, pr.J)
    for m in (70, 130):
        R = rng.standard_normal((pr.NV, m))
        X, its, rr = ctx1.shift_solve(-30.0, 1.0, R)
        assert rr.max() < 1e-10
        assert rel(X[:pr.NV], lu.solve(R)[:pr.NV]) < 1e-8


def test_nonconvergence_is_reported(cfg1):
    """An inner solve that cannot reach its tolerance is not an error (the ADI loop
    goes on, as the reference's loops do at *_max_steps) but must be visible."""
    pr, tb, trct, ms = cfg1
    F = (-pr.A - pr.Nc).tocsr()
    ctx = _lib.Context(0, gmres_maxit=5)
    ctx.set_operator(F.T.tocsr(), pr.M.T.tocsr(), pr.J)
    prm = _lib.adi_params(dict(adi_max_steps=2, project_w=False))
    with pytest.warns(RuntimeWarning, match="shift-solves stopped"):
        Z, info = ctx.lyap_adi(ms, trct, prm)
    assert info["gmres_nonconverged"] == 2 and info["gmres_worst_relres"] > 1e-11
    assert np.isfinite(Z).all()
    ctx.close()


def test_argument_errors(ctx1, cfg1):
    pr = cfg1[0]
    with pytest.raises(ValueError):
        ctx1.spmm(-1.0, 1.0, np.zeros((pr.NV, 4)))              # wrong row count
    with pytest.raises(ValueError):
        ctx1.spmm(-1.0, 1.0, np.zeros((pr.NV + pr.NP, 129)))    # too wide for one panel
    prm = _lib.adi_params(dict(adi_max_steps=3))
    with pytest.raises(ValueError):
        ctx1.lyap_adi([1.0], np.ones((pr.NV, 2)), prm)          # positive shift
    fresh = _lib.Context(0)
    with pytest.raises(RuntimeError):
        fresh.spmm(-1.0, 1.0, np.zeros((3, 1)))                 # operator not set
    fresh.close()


# --------------------------------------------------------------------- a2, a1
def test_lyap_adi_gain_vs_golden(ctx1, cfg1, golden):
    pr, tb, trct, ms = cfg1
    d = dict(pb.default_nwtn_adi_dict(), ms=ms)
    Z, info = ctx1.lyap_adi(ms, trct, _lib.adi_params(d))
    assert info["adi_steps"] == int(golden["lyap_steps"][0])
    K = -(pr.M.T @ (Z @ (Z.T @ tb.toarray())))
    assert rel(K, golden["K_lyap"]) < K_TOL
    # device gain kernel on the device-resident factor
    assert rel(-ctx1.gain(tb.toarray()), golden["K_lyap"]) < K_TOL


def test_newton_adi_gain_vs_golden_and_oracle(ctx1, cfg1, golden):
    pr, tb, trct, ms = cfg1
    d = dict(pb.default_nwtn_adi_dict(), ms=ms)
    Z, info = ctx1.ric_newtonadi(ms, tb.toarray(), trct, _lib.adi_params(d))
    assert info["nwtn_steps"] == int(golden["nwtn_steps"][0])
    # same Newton update norms as the oracle (same stopping decisions)
    assert np.isclose(info["upd_rel"], golden["upd_hist"][-1, 1], rtol=1e-3)
    K = -ctx1.gain(tb.toarray())
    assert rel(K, golden["K_ric"]) < K_TOL
    Kh = -opru.get_mTzzTtb(pr.M.T, Z, tb)      # from the factor copied to the host
    assert rel(Kh, golden["K_ric"]) < K_TOL
    # compression of the device-resident factor (solve_dae_ric.py:162-163 parameters)
    Zc, sv = ctx1.compress(None, thresh=5e-5, k=50)
    assert Zc.shape[1] == int(golden["kcomp"][0])
    assert np.allclose(sv[:20], golden["sv"][:20], rtol=1e-6)
    assert np.isclose(np.linalg.norm(Zc.T @ Zc), golden["gram_comp_fro"][0], rtol=1e-8)


def test_newton_with_initial_guess_and_old_feedback(ctx1, cfg1):
    """z0 / mtxoldb arguments (solve_dae_ric.py:152-159)."""
    pr, tb, trct, ms = cfg1
    F = (-pr.A - pr.Nc).tocsr()
    d = dict(pb.default_nwtn_adi_dict(), ms=ms, nwtn_max_steps=3)
    rng = np.random.default_rng(5)
    old = 1e-3 * olau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J,
                                         rhsv=rng.standard_normal((pr.NV, tb.shape[1])),
                                         transposedprj=True)
    z0 = 1e-2 * olau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=rng.standard_normal((pr.NV, 3)))
    ref = opru.proj_alg_ric_newtonadi(mmat=pr.M, amat=F, jmat=pr.J, bmat=tb, wmat=trct, z0=z0,
                                      mtxoldb=old, nwtn_adi_dict=d)
    Z, info = ctx1.ric_newtonadi(ms, tb.toarray(), trct, _lib.adi_params(d), Z0=z0, oldB=old)
    assert info["nwtn_steps"] == ref["nwtn_steps"]
    Kr = opru.get_mTzzTtb(pr.M.T, ref["zfac"], tb)
    assert rel(opru.get_mTzzTtb(pr.M.T, Z, tb), Kr) < K_TOL


# ----------------------------------------- the reference's own test, via the drop-in
def test_reference_unit_test_through_dropin():
    """The five identities of /root/reference/tests/test_units_compfacres_compress.py:85-106
    (tests/identities.py) with `pru` resolved to the MI355X drop-in package; set-up as there
    (:45-60): Stokes matrices, F = -M - 0.1 A - sparse random perturbation, W = randn(NV, 5),
    here seeded and with an explicit shift list."""
    import sadptprj_riclyap_adi.proj_ric_utils as pru
    from identities import check_reference_identities
    backend.reset()
    sm = pb.stokes_system(8, nu=1.0)
    M, A, J, NV = sm["M"], sm["A"], sm["J"], sm["NV"]
    rng = np.random.default_rng(0)
    pert = sps.random(NV, NV, density=0.03, format="csr", random_state=rng)
    F = (-M - 0.1 * A - 0.03 * M.diagonal().mean() * pert).tocsr()
    W = rng.standard_normal((NV, 5))
    d = dict(adi_max_steps=150, adi_newZ_reltol=1e-11, nwtn_max_steps=24, nwtn_upd_reltol=4e-7,
             nwtn_upd_abstol=4e-7, full_upd_norm_check=True, verbose=False,
             ms=pb.logshifts(2.0, 8e3, 12))
    Z, _ = check_reference_identities(pru, M, J, F, W, d)
    # the same Z as the oracle, as far as Z Z^T is concerned
    Zo = opru.solve_proj_lyap_stein(amat=F, mmat=M, jmat=J, wmat=W, adi_dict=d)["zfac"]
    assert opru.comp_diff_zzt_fnorm(Z, Zo) <= 1e-8 * np.linalg.norm(Zo.T @ Zo)
    # check_lyap_res (optcont_main.py:130) is honoured: the factored residual comes back
    out = pru.solve_proj_lyap_stein(amat=F, mmat=M, jmat=J, wmat=W, adi_dict=dict(d, check_lyap_res=True))
    assert 0.0 <= out["lyap_res"] < 1e-6 * np.linalg.norm(W.T @ W)
    backend.reset()


def test_unconverged_residual_norm_matches_oracle():
    import sadptprj_riclyap_adi.proj_ric_utils as pru
    backend.reset()
    sm = pb.stokes_system(6, nu=1.0)
    M, A, J, NV = sm["M"], sm["A"], sm["J"], sm["NV"]
    rng = np.random.default_rng(2)
    F = (-M - 0.1 * A).tocsr()
    W = rng.standard_normal((NV, 4))
    d = dict(adi_max_steps=3, adi_newZ_reltol=1e-30, ms=[-1.0, -4.0])
    Z = pru.solve_proj_lyap_stein(amat=F, mmat=M, jmat=J, wmat=W, adi_dict=d)["zfac"]
    Zo = opru.solve_proj_lyap_stein(amat=F, mmat=M, jmat=J, wmat=W, adi_dict=d)["zfac"]
    assert rel(Z, Zo) < 1e-8                    # sequential ADI: same columns, not just same ZZ^T
    r_gpu = pru.comp_proj_lyap_res_norm(Z, F, M, W, J)
    r_cpu = opru.comp_proj_lyap_res_norm(Zo, F, M, W, J)
    assert r_cpu > 0 and np.isclose(r_gpu, r_cpu, rtol=1e-7)
    backend.reset()


# ------------------------------------------------------------------ a3, a4, a6-a8
def test_compress_properties():
    import sadptprj_riclyap_adi.proj_ric_utils as pru
    backend.reset()
    rng = np.random.default_rng(9)
    n, c = 3000, 90
    Q, _ = np.linalg.qr(rng.standard_normal((n, c)))
    s = np.logspace(0, -9, c)
    Z = (Q * s) @ np.linalg.qr(rng.standard_normal((c, c)))[0]
    for thresh, k in ((1e-4, None), (None, 7), (1e-4, 5), (1e-30, None)):
        Zc = pru.compress_Zsvd(Z, thresh=thresh, k=k)
        Zo = opru.compress_Zsvd(Z, thresh=thresh, k=k)
        if thresh != 1e-30:
            assert Zc.shape == Zo.shape
        # truncation error of Z Z^T is that of the optimal rank-k' truncation
        kk = Zc.shape[1]
        tail = np.sqrt(np.sum(s[kk:] ** 4))
        err = opru.comp_diff_zzt_fnorm(Zc, Z)
        assert err <= 1.01 * tail + 1e-14
    # edge: a single column, and a zero threshold keeping everything numerically non-zero
    z1 = rng.standard_normal((n, 1))
    assert rel(pru.compress_Zsvd(z1, thresh=None, k=None) @ np.ones((1, 1)) * 1.0,
               z1 * np.sign((pru.compress_Zsvd(z1).T @ z1))) < 1e-12
    backend.reset()


def test_get_mtzzttb_sparse_and_dense(cfg1):
    import sadptprj_riclyap_adi.proj_ric_utils as pru
    backend.reset()
    pr, tb, trct, ms = cfg1
    rng = np.random.default_rng(1)
    Z = rng.standard_normal((pr.NV, 37))
    ref = opru.get_mTzzTtb(pr.M.T, Z, tb)
    assert rel(pru.get_mTzzTtb(pr.M.T, Z, tb), ref) < 1e-12            # sparse tb (optcont_main.py:505)
    f = rng.standard_normal((pr.NV, 1))
    assert rel(pru.get_mTzzTtb(pr.M.T, Z, f), opru.get_mTzzTtb(pr.M.T, Z, f)) < 1e-12   # dense NV x 1 (:183)
    backend.reset()


def test_gain_from_the_device_resident_factor(cfg1):
    """get_mTzzTtb right after the Newton iteration (optcont_main.py:488-506) uses the factor the iteration
    left on the device when it is handed the very array the iteration returned (made read-only for that);
    a copy of it goes through the upload path -- same K."""
    import sadptprj_riclyap_adi.proj_ric_utils as pru
    backend.reset()
    pr, tb, trct, ms = cfg1
    F = (-pr.A - pr.Nc).tocsr()
    d = dict(pb.default_nwtn_adi_dict(), ms=ms)
    out = pru.proj_alg_ric_newtonadi(mmat=pr.M, amat=F, jmat=pr.J, bmat=tb, wmat=trct, nwtn_adi_dict=d)
    Z = out["zfac"]
    assert not Z.flags.writeable
    ctx = backend.context()
    assert ctx._zdev is Z
    K_dev = pru.get_mTzzTtb(pr.M.T, Z, tb)
    K_up = pru.get_mTzzTtb(pr.M.T, Z.copy(), tb)
    assert rel(K_dev, K_up) < 1e-13
    assert rel(K_dev, opru.get_mTzzTtb(pr.M.T, Z, tb)) < 1e-12
    # any call that changes the device factor drops the shortcut
    pru.compress_Zsvd(Z, thresh=1e-9)
    assert ctx._zdev is None
    assert rel(pru.get_mTzzTtb(pr.M.T, Z, tb), K_up) < 1e-13
    backend.reset()


def test_lau_mirror(cfg1, golden):
    import sadptprj_riclyap_adi.lin_alg_utils as lau
    backend.reset()
    pr, tb, trct, ms = cfg1
    # a7: projection (optcont_main.py:405-408)
    got = lau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=pr.mc_mat.T, transposedprj=True)
    ref = olau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=pr.mc_mat.T, transposedprj=True)
    assert rel(got, ref) < 1e-8
    assert rel(lau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=trct),
               olau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=trct)) < 1e-8
    # a6: feed-forward saddle solve with the closed-loop low-rank term (optcont_main.py:510-514)
    wft = lau.solve_sadpnt_smw(amat=(pr.A + pr.Nc).T.tocsr(), jmat=pr.J, rhsv=golden["ff_rhs"],
                               umat=golden["K_ric"], vmat=tb.T)[:pr.NV]
    assert rel(wft, golden["ff_sol"]) < 1e-7
    # a8: mass inverse, NV-sized (solve_dae_ric.py:100) and small (optcont_main.py:398)
    assert rel(lau.apply_massinv(pr.M, trct), olau.apply_massinv(pr.M, trct)) < 1e-8
    got = lau.apply_massinv(pr.y_masmat, pr.mc_mat, output="sparse")
    assert sps.issparse(got)
    assert rel(got.toarray(), olau.apply_massinv(pr.y_masmat, pr.mc_mat)) < 1e-12
    backend.reset()


# ------------------------------------------------ properties at the benchmark size
def test_cfg2_size_properties():
    """N=58 (BASELINE cfg2: n = 29 930): residual, constraint and linearity checks
    that need no oracle factorisation, plus one LU cross-check."""
    pr = pb.ricc_problem(58, 0.05)
    calA = (-pr.A - pr.Nc).T.tocsr()
    ctx = _lib.Context(0)
    ctx.set_operator(calA, pr.M.T.tocsr(), pr.J)
    rng = np.random.default_rng(0)
    R = rng.standard_normal((pr.NV, 16))
    for p in (-1.0, -300.0):
        X, its, rr = ctx.shift_solve(p, 1.0, R)
        S = sps.bmat([[calA + p * pr.M, pr.J.T], [pr.J, None]], format="csr")
        res = S @ X - np.vstack([R, np.zeros((pr.NP, 16))])
        assert np.linalg.norm(res, axis=0).max() < 1e-10 * np.linalg.norm(R, axis=0).min()
        assert np.abs(pr.J @ X[:pr.NV]).max() < 1e-9 * np.abs(X).max()
    X2, _, _ = ctx.shift_solve(-300.0, 1.0, 2.0 * R[:, :4] - R[:, 4:8])
    assert rel(X2, 2.0 * X[:, :4] - X[:, 4:8]) < 1e-8
    Xo = olau.SaddleLU(calA - 300.0 * pr.M, pr.J).solve(R[:, :2])
    assert rel(X[:pr.NV, :2], Xo[:pr.NV]) < 1e-8
    ctx.close()


def test_shift_parallel_hipops_matches_sequential(cfg1, golden):
    """The multi-GPU orchestration on one GPU (world 1, sweep width 4)."""
    import torch
    from optconpy_amd.shift_parallel import HipOps, lyap_adi_shift_parallel
    pr, tb, trct, ms = cfg1
    F = (-pr.A - pr.Nc).tocsr()
    ctx = _lib.Context(0)
    ctx.set_operator(F.T.tocsr(), pr.M.T.tocsr(), pr.J)
    torch.cuda.set_device(0)
    ops = HipOps(ctx)
    W = ops.to_panel(trct)     # already projected (mct_mat_reg)
    blocks, info = lyap_adi_shift_parallel(ops, ms, W, adi_max_steps=200,
                                           adi_newZ_reltol=1e-8, width=4)
    Z = torch.cat(blocks, dim=1).cpu().numpy()
    K = -(pr.M.T @ (Z @ (Z.T @ tb.toarray())))
    assert rel(K, golden["K_lyap"]) < K_TOL
    assert info["width"] == 4 and ops.shift_solves == info["adi_steps"]
    # the same sweeps with two contexts (two HIP streams, two host threads)
    ctx2 = _lib.Context(0)
    ctx2.set_operator(F.T.tocsr(), pr.M.T.tocsr(), pr.J)
    ops2 = HipOps(ctx, [ctx2])
    blocks2, info2 = lyap_adi_shift_parallel(ops2, ms, W, adi_max_steps=200,
                                             adi_newZ_reltol=1e-8, width=4)
    Z2 = torch.cat(blocks2, dim=1).cpu().numpy()
    assert info2["adi_steps"] == info["adi_steps"] and ops2.shift_solves == info2["adi_steps"]
    assert opru.comp_diff_zzt_fnorm(Z2, Z) <= 1e-9 * np.linalg.norm(Z.T @ Z)
    ctx2.close()
    ctx.close()


def test_two_ranks_on_one_gpu():
    """The distributed shift-parallel path with device tensors: 2 processes on GPU 0
    (gloo; NCCL refuses two ranks per device), 2 streams each, all-gather + collective
    stop decision; both ranks must reproduce the golden gain."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "rehearse_2ranks.py")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    errs = [float(x) for x in re.findall(r"K rel diff vs golden ([0-9.e+-]+)", out.stdout)]
    solves = [int(x) for x in re.findall(r"local solves (\d+)", out.stdout)]
    steps = [int(x) for x in re.findall(r"(\d+) ADI steps", out.stdout)]
    assert len(errs) == 2 and max(errs) < K_TOL
    # the two ranks share the work (sweeps cut at the stopping step may leave one item of difference
    # per sweep); together they solved every ADI step once
    assert len(solves) == 2 and min(solves) > 0 and steps[0] == steps[1] <= sum(solves) <= steps[0] + 16


@pytest.mark.parametrize("n,c", [(300, 7), (3000, 70), (26450, 456)])
def test_tsqr_block_qr(n, c):
    """K5: Householder TSQR panels + block Gram-Schmidt: Z = Q R to rounding, Q orthonormal,
    R upper triangular -- also for a numerically rank-deficient Z (the Newton update norm
    factorises [Z_new, Z_old] with Z_new ~ Z_old)."""
    rng = np.random.default_rng(n + c)
    ctx = _lib.Context(0)
    ctx.set_dims(n)
    Z = rng.standard_normal((n, c))
    Q, R = ctx.qr(Z)
    assert np.allclose(np.tril(R, -1), 0.0)
    assert np.linalg.norm(Q @ R - Z) <= 1e-13 * np.linalg.norm(Z)
    assert np.linalg.norm(Q.T @ Q - np.eye(c)) <= 1e-12
    # |R| matches LAPACK's up to row signs
    Rl = np.linalg.qr(Z, mode="r")
    assert np.allclose(np.abs(np.diag(R)), np.abs(np.diag(Rl)), rtol=1e-10)
    # rank deficient: second half = first half + 1e-9 perturbation
    h = c // 2
    if h >= 2:
        Zd = np.hstack([Z[:, :h], Z[:, :h] + 1e-9 * rng.standard_normal((n, h))])
        Qd, Rd = ctx.qr(Zd)
        assert np.linalg.norm(Qd @ Rd - Zd) <= 1e-13 * np.linalg.norm(Zd)
        assert np.linalg.norm(Qd.T @ Qd - np.eye(2 * h)) <= 1e-10
        # || Z1 Z1^T - Z0 Z0^T ||_F from the factor, against the oracle's QR-based value
        S = np.r_[np.ones(h), -np.ones(h)]
        got = np.linalg.norm((Rd * S) @ Rd.T)
        ref = opru.comp_diff_zzt_fnorm(Zd[:, :h], Zd[:, h:])
        assert np.isclose(got, ref, rtol=1e-6)
        # an exactly zero column cannot go through the Cholesky-QR panels: the factorisation
        # falls back to the Householder TSQR tree and still delivers Z = Q R, Q^T Q = I
        Zz = Z[:, :2 * h].copy()
        Zz[:, 1] = 0.0
        Qz, Rz = ctx.qr(Zz)
        assert np.linalg.norm(Qz @ Rz - Zz) <= 1e-13 * np.linalg.norm(Zz)
        assert np.linalg.norm(Qz.T @ Qz - np.eye(2 * h)) <= 1e-12
        assert abs(Rz[1, 1]) <= 1e-14 * np.linalg.norm(Zz)
    ctx.close()


def test_compress_qr_mode_resolves_small_singular_values():
    """compress_qr=1: TSQR block QR + SVD of R.  Singular values far below
    sqrt(eps)*s_1 are resolved (the Gram route cannot), so the kept rank equals the
    oracle's QR + SVD (the reference's "QR ... SVD", optcont_main.py:133-134)."""
    rng = np.random.default_rng(21)
    n, c = 4000, 60
    Qm, _ = np.linalg.qr(rng.standard_normal((n, c)))
    s = np.logspace(0, -13, c)
    Z = (Qm * s) @ np.linalg.qr(rng.standard_normal((c, c)))[0]
    ctx = _lib.Context(0, compress_qr=1)
    ctx.set_dims(n)
    for thresh in (1e-6, 1e-10, 3e-12):
        Zc, sv = ctx.compress(Z, thresh=thresh)
        Zo = opru.compress_Zsvd(Z, thresh=thresh)
        assert Zc.shape == Zo.shape, (thresh, Zc.shape, Zo.shape)
        assert np.allclose(sv[:Zc.shape[1]], s[:Zc.shape[1]], rtol=1e-6)
        assert opru.comp_diff_zzt_fnorm(Zc, Zo) <= 1e-12 * np.linalg.norm(Zo.T @ Zo)
    ctx.close()
    # the default Gram route: same result down to sqrt(eps), not below
    ctx = _lib.Context(0)
    ctx.set_dims(n)
    Zc, _ = ctx.compress(Z, thresh=1e-6)
    assert Zc.shape == opru.compress_Zsvd(Z, thresh=1e-6).shape
    ctx.close()


def test_newton_parity_other_ordering_and_viscosity():
    """Interleaved dof ordering and nu = 0.02 (four Newton steps): same step count as the
    oracle and K within the parity bar."""
    for nu, order in ((0.02, "component"), (0.1, "interleaved")):
        pr = pb.ricc_problem(15, nu, ordering=order)
        F = (-pr.A - pr.Nc).tocsr()
        mct = olau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=pr.mc_mat.T, transposedprj=True)
        tb = olau.apply_invsqrt_fromright(pr.rmat, pr.b_mat, output="dense")
        trct = olau.apply_invsqrt_fromright(pr.y_masmat, mct, output="dense")
        ms = pb.logshifts(1.0, 1e3, 8)
        d = dict(pb.default_nwtn_adi_dict(), ms=ms)
        ctx = _lib.Context(0)
        ctx.set_operator(F.T.tocsr(), pr.M.T.tocsr(), pr.J)
        Z, info = ctx.ric_newtonadi(ms, tb, trct, _lib.adi_params(d))
        K = -ctx.gain(tb)
        ctx.close()
        ref = opru.proj_alg_ric_newtonadi(mmat=pr.M, amat=F, jmat=pr.J, bmat=tb, wmat=trct,
                                          nwtn_adi_dict=d)
        assert info["nwtn_steps"] == ref["nwtn_steps"]
        assert rel(K, -opru.get_mTzzTtb(pr.M.T, ref["zfac"], tb)) < K_TOL


def test_batched_shift_solve_matches_single():
    """ricadi_shift_solve_batch_dev: all shifts of a sweep in one lockstep launch sequence give
    the same solutions as one solve per shift -- shared and per-group right-hand sides, with
    the low-rank term.  (Iteration counts are of the same order but not equal: the restart-cycle
    length adapts to the slowest group of the batch.)"""
    import torch
    pr = pb.ricc_problem(15, 0.05)
    ctx = _lib.Context(0)
    ctx.set_operator((-pr.A - pr.Nc).T.tocsr(), pr.M.T.tocsr(), pr.J)
    rng = np.random.default_rng(5)
    ctx.set_lowrank(0.1 * rng.standard_normal((pr.NV, 3)), rng.standard_normal((pr.NV, 3)))
    m = 7
    ps = [-0.7, -3.0, -40.0, -900.0, -12.0]
    dev = torch.device("cuda", 0)
    for shared in (True, False):
        R = rng.standard_normal((1 if shared else len(ps), pr.NV, m))
        Rd = torch.as_tensor(R).to(dev)
        X = torch.empty(len(ps), ctx.n, m, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        its, rr = ctx.shift_solve_batch_dev(ps, [1.0] * len(ps), Rd.data_ptr(),
                                            0 if shared else pr.NV * m, m, X.data_ptr())
        ctx.synchronize()
        assert rr.max() < 1e-9
        Xh = X.cpu().numpy()
        for g, p in enumerate(ps):
            Xs, it1, rr1 = ctx.shift_solve(p, 1.0, R[0 if shared else g])
            assert abs(it1 - its[g]) <= max(2, 0.75 * max(it1, its[g]))
            assert rel(Xh[g], Xs) < 1e-8
    ctx.close()


@pytest.mark.parametrize("m", [1, 5, 7])
def test_odd_panel_widths_through_adi_and_batched_solve(m):
    """Regression for the round-1 abort (N = 8 Stokes pencil, m = 5, adi_max_steps = 150, more
    factor columns than rows): panel widths that are no multiple of 4 through the step-wise
    ADI, the sweep form and the batched solve, against the oracle."""
    import torch
    sm = pb.stokes_system(8, nu=1.0)
    M, A, J, NV = sm["M"], sm["A"], sm["J"], sm["NV"]
    F = (-M - 0.1 * A).tocsr()
    rng = np.random.default_rng(m)
    W = rng.standard_normal((NV, m))
    ms = pb.logshifts(2.0, 8e3, 12)
    d = dict(adi_max_steps=150, adi_newZ_reltol=1e-11, ms=ms)
    ctx = _lib.Context(0)
    ctx.set_operator(F.T.tocsr(), M.T.tocsr(), J)
    Zo = opru.solve_proj_lyap_stein(amat=F, mmat=M, jmat=J, wmat=W, adi_dict=d)["zfac"]
    Z1, i1 = ctx.lyap_adi(ms, W, _lib.adi_params(d))
    assert i1["cols"] == Z1.shape[1] == i1["adi_steps"] * m
    assert opru.comp_diff_zzt_fnorm(Z1, Zo) <= 1e-8 * np.linalg.norm(Zo.T @ Zo)
    Z2, i2 = ctx.lyap_adi(ms, W, _lib.adi_params(dict(d, sweep_width=4)))
    assert opru.comp_diff_zzt_fnorm(Z2, Zo) <= 1e-7 * np.linalg.norm(Zo.T @ Zo)
    # in-ADI recompression with a factor that has more columns than rows
    Z3, i3 = ctx.lyap_adi(ms, W, _lib.adi_params(dict(d, compress_cols=64)))
    assert Z3.shape[1] <= NV and opru.comp_diff_zzt_fnorm(Z3, Zo) <= 1e-7 * np.linalg.norm(Zo.T @ Zo)
    dev = torch.device("cuda", 0)
    ps = [-2.0, -50.0, -3000.0]
    Rd = torch.as_tensor(W).to(dev)
    X = torch.empty(len(ps), ctx.n, m, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    its, rr = ctx.shift_solve_batch_dev(ps, [1.0] * len(ps), Rd.data_ptr(), 0, m, X.data_ptr())
    ctx.synchronize()
    Xh = X.cpu().numpy()
    for g, p in enumerate(ps):
        Xo = olau.SaddleLU(F.T.tocsr() + p * M.T.tocsr(), J).solve(W)
        assert rel(Xh[g], Xo) < 1e-8
    ctx.close()


def test_cpp_sweep_adi_matches_stepwise_adi():
    """ricadi_lyap_adi with sweep_width = G (G steps = one batched solve + Cauchy recombination)
    against the step-by-step recurrence: same Z Z^T after the same number of steps; a shift
    list with repeats silently falls back to the step-by-step form."""
    pr = pb.ricc_problem(15, 0.05)
    ctx = _lib.Context(0)
    ctx.set_operator((-pr.A - pr.Nc).T.tocsr(), pr.M.T.tocsr(), pr.J)
    rng = np.random.default_rng(11)
    W = rng.standard_normal((pr.NV, 5))
    B = rng.standard_normal((pr.NV, 3))
    ms = pb.logshifts(1.0, 1e3, 8)
    for G in (2, 4, 8):
        d = dict(adi_max_steps=24, adi_newZ_reltol=0.0, ms=ms)
        Z1, i1 = ctx.lyap_adi(ms, W, _lib.adi_params(d))
        Zs, i2 = ctx.lyap_adi(ms, W, _lib.adi_params(dict(d, sweep_width=G)))
        assert i1["adi_steps"] == i2["adi_steps"] == 24 and Zs.shape == Z1.shape
        assert rel(Zs @ (Zs.T @ B), Z1 @ (Z1.T @ B)) < 1e-8
    # stopping rule at sweep granularity; the last sweeps are narrowed (2 / 4 / 8 steps) when the
    # decay of the block norms predicts that fewer than a full sweep is still needed
    d = dict(adi_max_steps=200, adi_newZ_reltol=1e-8, ms=ms)
    Z1, i1 = ctx.lyap_adi(ms, W, _lib.adi_params(d))
    Zs, i2 = ctx.lyap_adi(ms, W, _lib.adi_params(dict(d, sweep_width=8)))
    assert i2["adi_rel_newZ"] < 1e-8 and i2["adi_steps"] == i1["adi_steps"]     # the reference's rule, step for step
    # a step budget that is no multiple of the sweep width is used up to the last step
    Zp, i3 = ctx.lyap_adi(ms, W, _lib.adi_params(dict(adi_max_steps=21, adi_newZ_reltol=0.0, ms=ms, sweep_width=8)))
    Zq, i4 = ctx.lyap_adi(ms, W, _lib.adi_params(dict(adi_max_steps=21, adi_newZ_reltol=0.0, ms=ms)))
    assert i3["adi_steps"] == i4["adi_steps"] == 21 and rel(Zp @ (Zp.T @ B), Zq @ (Zq.T @ B)) < 1e-8
    assert rel(Zs @ (Zs.T @ B), Z1 @ (Z1.T @ B)) < 1e-6
    # repeated shifts: no sweeps possible, the result is the step-by-step one
    ms_rep = np.r_[ms[:4], ms[:4]]
    d = dict(adi_max_steps=16, adi_newZ_reltol=0.0)
    Za, _ = ctx.lyap_adi(ms_rep, W, _lib.adi_params(d))
    Zb, _ = ctx.lyap_adi(ms_rep, W, _lib.adi_params(dict(d, sweep_width=4)))
    assert rel(Zb, Za) < 1e-7
    ctx.close()


def test_sweep_of_the_whole_16_shift_cycle_matches_oracle():
    """The benchmark's configuration in small: 16 log-spaced shifts over 3.5 decades swept in
    ONE batch (16 x 16 Cauchy matrix), through HipOps (batched solve + device recombination),
    against the oracle's step-by-step ADI with the same shifts."""
    import torch
    from optconpy_amd.shift_parallel import HipOps, lyap_adi_shift_parallel
    pr = pb.ricc_problem(15, 0.05)
    F = (-pr.A - pr.Nc).tocsr()
    mct = olau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=pr.mc_mat.T, transposedprj=True)
    tb = olau.apply_invsqrt_fromright(pr.rmat, pr.b_mat, output="dense")
    trct = olau.apply_invsqrt_fromright(pr.y_masmat, mct, output="dense")
    ms = pb.logshifts(1.0, 3e3, 16)
    ctx = _lib.Context(0)
    ctx.set_operator(F.T.tocsr(), pr.M.T.tocsr(), pr.J)
    torch.cuda.set_device(0)
    ops = HipOps(ctx)
    blocks, info = lyap_adi_shift_parallel(ops, ms, ops.to_panel(trct), adi_max_steps=200,
                                           adi_newZ_reltol=1e-9, width=16)
    Z = torch.cat(blocks, dim=1).cpu().numpy()
    ctx.close()
    ref = opru.solve_proj_lyap_stein(amat=F, mmat=pr.M, jmat=pr.J, wmat=trct,
                                     adi_dict=dict(adi_max_steps=200, adi_newZ_reltol=1e-9, ms=ms))
    # the sweep form ends after the same step as the oracle's step-by-step iteration
    assert info["width"] == 16 and info["adi_steps"] == ref["adi_steps"]
    K = pr.M.T @ (Z @ (Z.T @ tb))
    Ko = opru.get_mTzzTtb(pr.M.T, ref["zfac"], tb)
    assert rel(K, Ko) < K_TOL
