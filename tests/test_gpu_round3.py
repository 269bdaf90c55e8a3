"""GPU tests of round 3: recycled initial guesses, the truncated-sweep residual factor, the storage
safety net of the inner GMRES, and the rank-sharded sweeps inside the library (two ranks on one GPU).

All of them go through the drop-in boundary (``sadptprj_riclyap_adi`` -> C-ABI) and compare with the
oracle (CPU restatement) or with identities recomputed on the host.
"""
import json
import os
import subprocess
import sys
import warnings

import numpy as np
import pytest

from optconpy_amd import backend, problems as pb
from oracle import lin_alg_utils as olau, proj_ric_utils as opru

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _inputs(N, nu, nshifts=8, pmax=1e3):
    import sadptprj_riclyap_adi.lin_alg_utils as lau
    pr = pb.ricc_problem(N, nu, NU=4, NY=4, alphau=1e-2)
    mct = lau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=pr.mc_mat.T, transposedprj=True)
    tb = lau.apply_invsqrt_fromright(pr.rmat, pr.b_mat, output="dense")
    trct = lau.apply_invsqrt_fromright(pr.y_masmat, mct, output="dense")
    return pr, tb, trct, pb.logshifts(1.0, pmax, nshifts)


def test_recycled_guesses_same_gain_fewer_iterations(monkeypatch):
    """ricadi_set_recycle / RICADI_RECYCLE: the sweeps of the Newton-ADI start from the least-squares
    combination of their last three solved right-hand sides.  Same K as without (and as the oracle),
    same ADI / Newton step counts, fewer GMRES iterations."""
    import sadptprj_riclyap_adi.proj_ric_utils as pru
    backend.reset()
    pr, tb, trct, ms = _inputs(20, 0.05, nshifts=8)
    F = (-pr.A - pr.Nc).tocsr()
    d = dict(pb.default_nwtn_adi_dict(), ms=ms, sweep_width=8)
    ref = opru.proj_alg_ric_newtonadi(mmat=pr.M, amat=F, jmat=pr.J, bmat=tb, wmat=trct, nwtn_adi_dict=dict(d))
    K_o = -opru.get_mTzzTtb(pr.M.T, ref["zfac"], tb)
    res = {}
    for depth in ("0", "3"):
        monkeypatch.setenv("RICADI_RECYCLE", depth)
        backend.reset()
        out = pru.proj_alg_ric_newtonadi(mmat=pr.M, amat=F, jmat=pr.J, bmat=tb, wmat=trct, nwtn_adi_dict=d)
        assert out["gmres_nonconverged"] == 0
        res[depth] = (out, -pru.get_mTzzTtb(pr.M.T, out["zfac"], tb))
    (o0, K0), (o3, K3) = res["0"], res["3"]
    assert rel(K0, K_o) < 1e-6 and rel(K3, K_o) < 1e-6
    assert rel(K3, K0) < 1e-8
    assert o3["nwtn_steps"] == o0["nwtn_steps"] == ref["nwtn_steps"]
    assert o3["adi_steps"] == o0["adi_steps"]
    assert o3["gmres_iters"] < 0.97 * o0["gmres_iters"], (o3["gmres_iters"], o0["gmres_iters"])
    backend.reset()


def test_truncated_sweep_residual_is_the_residual_of_the_returned_factor():
    """Sweep form with a stop in the middle of the first sweeps (blocks really dropped): the residual
    norm the ADI reports (||W_end^T W_end||_F) equals the factored residual of the factor it returns,
    evaluated independently (a5) -- on the GPU and against the oracle's evaluation."""
    import sadptprj_riclyap_adi.proj_ric_utils as pru
    backend.reset()
    pr, tb, trct, ms = _inputs(12, 0.1, nshifts=8, pmax=500.0)
    F = (-pr.A - pr.Nc).tocsr()
    hit = None
    for tol in (0.5, 0.3, 0.2, 0.1, 0.05, 0.02, 0.01):
        out = pru.solve_proj_lyap_stein(amat=F, mmat=pr.M, jmat=pr.J, wmat=trct,
                                        adi_dict=dict(adi_max_steps=40, adi_newZ_reltol=tol, ms=ms, sweep_width=8))
        if out["adi_steps"] % 8 != 0 and out["adi_steps"] < 16:
            hit = out
            break
    assert hit is not None, "no tolerance found that stops inside one of the first two sweeps"
    Z = hit["zfac"]
    assert Z.shape[1] == hit["adi_steps"] * trct.shape[1]
    r_gpu = np.sqrt(pru.comp_proj_lyap_res_norm(Z, F, pr.M, trct, pr.J))
    r_orc = np.sqrt(opru.comp_proj_lyap_res_norm(Z, F, pr.M, trct, pr.J))
    assert np.isclose(r_gpu, r_orc, rtol=1e-6)
    assert np.isclose(hit["res_fro"], r_orc, rtol=1e-5), (hit["res_fro"], r_orc)
    backend.reset()


def test_hard_operator_converges_without_warning_and_escalates_when_starved():
    """The operator round 2's probe flagged (N = 15, nu = 0.005, p = -1: ~450 iterations with the
    FP16-stored basis): (i) with the default options the solve converges, no RuntimeWarning;
    (ii) starved of iterations (gmres_maxit = 200) the group is continued with the FP32- and then the
    FP64-stored basis and FP64 preconditioner inverses instead of being folded into Z unconverged:
    it converges, and the escalations are reported."""
    import sadptprj_riclyap_adi.proj_ric_utils as pru
    pr, tb, trct, _ = _inputs(15, 0.005)
    F = (-pr.A - pr.Nc).tocsr()
    d = dict(adi_max_steps=1, adi_newZ_reltol=0.0, ms=[-1.0], sweep_width=1)
    backend.reset()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        a = pru.solve_proj_lyap_stein(amat=F, mmat=pr.M, jmat=pr.J, wmat=trct, adi_dict=d)
    assert a["gmres_nonconverged"] == 0 and a["storage_escalations"] == 0
    backend.configure(gmres_maxit=200)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            b = pru.solve_proj_lyap_stein(amat=F, mmat=pr.M, jmat=pr.J, wmat=trct, adi_dict=d)
        assert b["gmres_nonconverged"] == 0
        assert b["storage_escalations"] >= 1
        assert b["gmres_iters"] > 200
        assert rel(b["zfac"], a["zfac"]) < 1e-7
    finally:
        backend.configure()
    # the same single step by the oracle
    o = opru.solve_proj_lyap_stein(amat=F, mmat=pr.M, jmat=pr.J, wmat=trct, adi_dict=d)
    assert rel(a["zfac"] @ a["zfac"].T, o["zfac"] @ o["zfac"].T) < 1e-7


def test_two_ranks_full_newton_step_through_the_library_exchange():
    """bench.py --gpus 2 without WORLD_SIZE starts its own two ranks (child torchrun); with
    --rehearse-one-gpu both sit on this box's one GPU and the all-gather goes through gloo.  The step
    is the N = 1 step (projection, cut sweeps, update norm, recompression, gain) with the sweeps sharded
    by shift inside the library: K matches the oracle fixture, 177 shift-solves as on one GPU."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu",
                        "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["K_rel_diff_vs_oracle"] < 1e-6
    assert d["config"]["shift_solves_per_step"] == 177
    assert "PARITY_VIOLATION" not in d["config"]


def _graded_factor(nv, c, decades, seed):
    """nv x c factor with singular values falling over `decades` orders of magnitude, mixed into every
    column (the shape of a low-rank ADI factor: far more columns than numerical rank)."""
    rng = np.random.default_rng(seed)
    r = min(nv, c)
    Q, _ = np.linalg.qr(rng.standard_normal((nv, r)))
    V, _ = np.linalg.qr(rng.standard_normal((c, r)))
    return (Q * np.logspace(0, -decades, r)) @ V.T


@pytest.mark.parametrize("nv,c,decades", [(3000, 700, 14), (2500, 1500, 20), (900, 40, 3), (400, 600, 12)])
def test_recompress_pivoted_cholesky_route(nv, c, decades):
    """ricadi_recompress (the drivers' internal recompression since round 3: pivoted Cholesky of the Gram
    matrix + orthonormalised factor rows, no eigensolver): Zc Zc^T = Z Z^T to rounding, the column count
    that of the optimal truncation at the same level (sqrt(eps) sigma_1) up to a few columns, and
    range(Zc) inside range(Z) (the reference's compression identity, tests/...compress.py:92-97)."""
    from optconpy_amd import _lib
    Z = _graded_factor(nv, c, decades, seed=c)
    ctx = _lib.Context(0)
    ctx.set_dims(nv)
    Zc = ctx.recompress(Z)
    ctx.close()
    s = np.linalg.svd(Z, compute_uv=False)
    k_opt = int((s > 3e-8 * s[0]).sum())
    X = Z @ Z.T
    assert np.linalg.norm(Zc @ Zc.T - X) <= 2e-14 * np.linalg.norm(X)
    assert k_opt - 2 <= Zc.shape[1] <= k_opt + max(6, k_opt // 20), (Zc.shape[1], k_opt)
    # range(Zc) in range(Z)
    U = np.linalg.svd(Z, full_matrices=False)[0][:, :min(k_opt + 40, min(Z.shape))]
    assert np.linalg.norm(Zc - U @ (U.T @ Zc)) <= 1e-7 * np.linalg.norm(Zc)


def test_device_resident_boundary_same_factor_and_gain():
    """The mirror's device-resident form of the boundary calls (ricadi_ric_newtonadi_dev, DeviceFactor): panels
    staged in HBM once, z0 and the new factor never cross PCIe.  Same Newton steps, same ADI steps, same K as the
    ndarray form of the same calls and as the oracle (optcont_main.py:488-492,505; solve_dae_ric.py:152-159 with
    z0 = the previous factor)."""
    import sadptprj_riclyap_adi.proj_ric_utils as pru
    backend.reset()
    pr, tb, trct, ms = _inputs(15, 0.1, nshifts=8)
    F = (-pr.A - pr.Nc).tocsr()
    MT = pr.M.T.tocsr()
    d = dict(pb.default_nwtn_adi_dict(), ms=ms, sweep_width=8)
    host = pru.proj_alg_ric_newtonadi(mmat=pr.M, amat=F, jmat=pr.J, bmat=tb, wmat=trct, nwtn_adi_dict=d)
    K_h = -pru.get_mTzzTtb(MT, host["zfac"], tb)
    tb_d, trct_d = pru.to_device(tb), pru.to_device(trct)
    dev = pru.proj_alg_ric_newtonadi(mmat=pr.M, amat=F, jmat=pr.J, bmat=tb_d, wmat=trct_d, nwtn_adi_dict=d)
    assert isinstance(dev["zfac"], pru.DeviceFactor) and dev["zfac"].shape == host["zfac"].shape
    assert dev["nwtn_steps"] == host["nwtn_steps"] and dev["adi_steps"] == host["adi_steps"]
    K_d = -pru.get_mTzzTtb(MT, dev["zfac"], tb_d)
    assert rel(K_d, K_h) < 1e-9
    ref = opru.proj_alg_ric_newtonadi(mmat=pr.M, amat=F, jmat=pr.J, bmat=tb, wmat=trct, nwtn_adi_dict=dict(d))
    assert rel(K_d, -opru.get_mTzzTtb(pr.M.T, ref["zfac"], tb)) < 1e-6
    # one more step from the device-resident iterate (z0 a DeviceFactor) == from its host copy
    d1 = dict(d, nwtn_max_steps=1)
    nxt_d = pru.proj_alg_ric_newtonadi(mmat=pr.M, amat=F, jmat=pr.J, bmat=tb_d, wmat=trct_d, z0=dev["zfac"],
                                       nwtn_adi_dict=d1)
    nxt_h = pru.proj_alg_ric_newtonadi(mmat=pr.M, amat=F, jmat=pr.J, bmat=tb, wmat=trct, z0=np.asarray(dev["zfac"]),
                                       nwtn_adi_dict=d1)
    assert nxt_d["adi_steps"] == nxt_h["adi_steps"]
    assert rel(-pru.get_mTzzTtb(MT, nxt_d["zfac"], tb_d), -pru.get_mTzzTtb(MT, nxt_h["zfac"], tb)) < 1e-9
    # a DeviceFactor is accepted wherever the mirror takes an ndarray (downloaded on demand)
    Zc = pru.compress_Zsvd(dev["zfac"], thresh=1e-7)
    assert Zc.shape[0] == pr.NV and Zc.shape[1] <= dev["zfac"].shape[1]
    backend.reset()


def test_smoothed_aggregation_fewer_iterations_same_solutions(monkeypatch):
    """Round 3: the velocity prolongation of the two-level preconditioner is smoothed, P = (I - w D^-1 sym(cal A)) Y
    (shift independent; (P - Y) e rides in the first velocity sweep).  Same solutions to the GMRES tolerance,
    clearly fewer iterations on the stiffness-dominated shifts; switched off by itself for a mass-like operator
    (lau.app_prj_via_sadpnt hands the mass matrix over as cal A), where it would hurt."""
    import torch
    import scipy.sparse as sps
    from optconpy_amd import _lib
    pr = pb.ricc_problem(30, 0.05)
    calA, calE = (-pr.A - pr.Nc).T.tocsr(), pr.M.T.tocsr()
    ms = [float(p) for p in pb.logshifts(1.0, 1e3, 8)]
    R = np.random.default_rng(11).standard_normal((pr.NV, 16))
    dev = torch.device("cuda", 0)
    Rd = torch.as_tensor(R).to(dev)

    def solve(A, E, shifts, betas, sa):
        monkeypatch.setenv("RICADI_SA", sa)
        ctx = _lib.Context(0)
        ctx.set_operator(A, E, pr.J)
        X = torch.empty(len(shifts), ctx.n, 16, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        its, rr = ctx.shift_solve_batch_dev(shifts, betas, Rd.data_ptr(), 0, 16, X.data_ptr())
        ctx.synchronize()
        Xh = X.cpu().numpy()
        ctx.close()
        assert rr.max() <= 1e-10 * 1.0000001
        return Xh, its

    X0, it0 = solve(calA, calE, ms, [1.0] * len(ms), "0")
    X1, it1 = solve(calA, calE, ms, [1.0] * len(ms), "0.5")
    for g in range(len(ms)):
        assert rel(X1[g][:pr.NV], X0[g][:pr.NV]) < 1e-8
    assert sum(it1) < 0.9 * sum(it0) and it1[0] < 0.85 * it0[0], (it0, it1)
    # mass-like operator: same iteration counts with and without the switch (the criterion keeps it off)
    zero = sps.csr_matrix(calE.shape)
    _, im0 = solve(calE, zero, [0.0], [1.0], "0")
    _, im1 = solve(calE, zero, [0.0], [1.0], "0.5")
    assert im0 == im1, (im0, im1)
