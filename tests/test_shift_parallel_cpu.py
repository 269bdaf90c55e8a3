"""Shift-parallel ADI host logic on CPU: in-process (G = 4) and 2 ranks over gloo.

The local solves are injected from the oracle (SuperLU); what is under test is
the product's orchestration: shift dealing, all-gather layout, Cauchy
recombination, residual hand-off, stopping rule.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from optconpy_amd import problems as pb
from optconpy_amd.shift_parallel import item_layout, lyap_adi_shift_parallel, plan_items, sweep_shifts
from oracle import lin_alg_utils as olau, proj_ric_utils as opru


class OracleOps:
    """CPU stand-in for HipOps: same interface, oracle arithmetic."""

    def __init__(self, calA, calE, J):
        self.calA, self.calE, self.J = calA, calE, J
        self.lus = {}
        self.nv = calA.shape[0]

    def solve(self, p, W):
        if p not in self.lus:
            self.lus[p] = olau.SaddleLU(self.calA + p * self.calE, self.J)
        return torch.from_numpy(self.lus[p].solve(W.numpy())[:self.nv].copy())

    def lincomb(self, coef, U_all):
        return torch.einsum("i,inm->nm", torch.as_tensor(np.asarray(coef)), U_all).contiguous()

    def apply_E(self, coef, V, W):
        W += torch.from_numpy(coef * (self.calE @ V.numpy()))

    def fro2(self, T):
        return float((T * T).sum())

    def gram_fro(self, T):
        return float(torch.linalg.norm(T.T @ T))


def _problem(N=6):
    pr = pb.ricc_problem(N, 0.1, NU=2, NY=2)
    F = (-pr.A - pr.Nc).tocsr()
    mct = olau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=pr.mc_mat.T, transposedprj=True)
    W = olau.apply_invsqrt_fromright(pr.y_masmat, mct, output="dense")
    tb = olau.apply_invsqrt_fromright(pr.rmat, pr.b_mat, output="dense")
    return pr, F, W, tb


def test_sweep_shift_dealing():
    ms = [-1.0, -2.0, -3.0, -4.0, -5.0, -6.0, -7.0]
    assert sweep_shifts(ms, 0, 4) == [-1.0, -2.0, -3.0, -4.0]
    assert sweep_shifts(ms, 1, 4) == [-5.0, -6.0, -7.0, -1.0]   # cycles like the sequential loop


@pytest.mark.parametrize("G", [1, 2, 4, 8])
def test_blocked_equals_sequential_fixed_steps(G):
    """G shifts per sweep, no early stop: X equals sequential LR-ADI to rounding."""
    pr, F, W, tb = _problem()
    ms = pb.logshifts(1.0, 500.0, 8)
    steps = 16
    d = dict(adi_max_steps=steps, adi_newZ_reltol=0.0, ms=ms)
    Zs = opru.solve_proj_lyap_stein(amat=F, mmat=pr.M, jmat=pr.J, wmat=W, adi_dict=d)
    ops = OracleOps(F.T.tocsr(), pr.M.T.tocsr(), pr.J)
    blocks, info = lyap_adi_shift_parallel(ops, ms, torch.from_numpy(W.copy()),
                                           adi_max_steps=steps, adi_newZ_reltol=0.0, width=G)
    Zb = torch.cat(blocks, dim=1).numpy()
    assert info["adi_steps"] == steps and Zb.shape == Zs["zfac"].shape
    Xs = Zs["zfac"] @ Zs["zfac"].T
    assert np.linalg.norm(Zb @ Zb.T - Xs) <= 1e-11 * np.linalg.norm(Xs)
    # residual factor hand-off: same Gram matrix as the sequential residual
    assert np.isclose(info["res_fro"], Zs["res_hist"][-1], rtol=1e-8, atol=1e-18)
    Ks = opru.get_mTzzTtb(pr.M.T, Zs["zfac"], tb)
    Kb = opru.get_mTzzTtb(pr.M.T, Zb, tb)
    assert np.linalg.norm(Kb - Ks) <= 1e-11 * np.linalg.norm(Ks)


def test_item_layout_and_plan():
    """Work items (shift, column part) of a sweep: every item has exactly one owner and one
    position in the rank-major gathered buffer; ranks differ by at most one item."""
    assert plan_items(16, 1) == 1 and plan_items(16, 8) == 1       # shifts cover the ranks
    assert plan_items(4, 8) == 2 and plan_items(1, 2) == 2         # more ranks than shifts
    assert plan_items(16, 8, col_parts=2) == 2
    for G, parts, world in ((16, 1, 8), (16, 2, 8), (8, 2, 3), (4, 2, 8), (5, 1, 2)):
        items, per_rank = item_layout(G, parts, world)
        assert len(items) == G * parts and per_rank == -(-G * parts // world)
        assert sorted((it["g"], it["q"]) for it in items) == [(g, q) for g in range(G) for q in range(parts)]
        pos = [it["pos"] for it in items]
        assert len(set(pos)) == len(pos) and max(pos) < world * per_rank
        for it in items:
            assert it["pos"] == it["rank"] * per_rank + it["slot"] and it["slot"] < per_rank
        load = np.bincount([it["rank"] for it in items], minlength=world)
        assert load.max() - load.min() <= 1
    # fixed owners (the library's dealing): every item with its owner, slots counted per rank
    from optconpy_amd import _lib
    ms16 = pb.logshifts(1.0, 3e3, 16)
    for world in (2, 4, 8):
        own = _lib.host_deal(ms16, world)
        items, per_rank = item_layout(16, 1, world, own)
        assert [it["rank"] for it in items] == [int(o) for o in own]
        assert per_rank == np.bincount(own, minlength=world).max()
        pos = [it["pos"] for it in items]
        assert len(set(pos)) == 16 and max(pos) < world * per_rank


@pytest.mark.parametrize("G,parts", [(4, 2), (8, 4), (2, 1)])
def test_column_split_equals_sequential(G, parts):
    """Column parts of the residual factor solved as separate work items (SURVEY.md 8e,
    "alternative axis"): same X as sequential LR-ADI."""
    pr, F, W, tb = _problem()
    assert W.shape[1] % parts == 0
    ms = pb.logshifts(1.0, 500.0, 8)
    steps = 16
    d = dict(adi_max_steps=steps, adi_newZ_reltol=0.0, ms=ms)
    Zs = opru.solve_proj_lyap_stein(amat=F, mmat=pr.M, jmat=pr.J, wmat=W, adi_dict=d)
    ops = OracleOps(F.T.tocsr(), pr.M.T.tocsr(), pr.J)
    blocks, info = lyap_adi_shift_parallel(ops, ms, torch.from_numpy(W.copy()), adi_max_steps=steps,
                                           adi_newZ_reltol=0.0, width=G, col_parts=parts)
    Zb = torch.cat(blocks, dim=1).numpy()
    assert info["adi_steps"] == steps and info["col_parts"] == parts and Zb.shape == Zs["zfac"].shape
    Xs = Zs["zfac"] @ Zs["zfac"].T
    assert np.linalg.norm(Zb @ Zb.T - Xs) <= 1e-11 * np.linalg.norm(Xs)
    assert np.isclose(info["res_fro"], Zs["res_hist"][-1], rtol=1e-8, atol=1e-18)


@pytest.mark.parametrize("G,parts", [(4, 1), (8, 2)])
def test_stops_after_the_same_step_as_the_sequential_iteration(G, parts):
    """The blocks of a sweep are the blocks of the step-by-step iteration, so the reference's
    rule (relative norm of the new block, optcont_main.py:123-124) ends the sweep form after
    exactly the oracle's step; the sweep containing that step is cut short by prediction or
    its surplus blocks are dropped."""
    pr, F, W, tb = _problem()
    ms = pb.logshifts(1.0, 500.0, 8)
    for tol in (1e-6, 1e-9):
        d = dict(adi_max_steps=200, adi_newZ_reltol=tol, ms=ms)
        Zs = opru.solve_proj_lyap_stein(amat=F, mmat=pr.M, jmat=pr.J, wmat=W, adi_dict=d)
        ops = OracleOps(F.T.tocsr(), pr.M.T.tocsr(), pr.J)
        blocks, info = lyap_adi_shift_parallel(ops, ms, torch.from_numpy(W.copy()), adi_max_steps=200,
                                               adi_newZ_reltol=tol, width=G, col_parts=parts)
        Zb = torch.cat(blocks, dim=1).numpy()
        assert info["adi_steps"] == Zs["adi_steps"] and Zb.shape == Zs["zfac"].shape
        Xs = Zs["zfac"] @ Zs["zfac"].T
        assert np.linalg.norm(Zb @ Zb.T - Xs) <= 1e-10 * np.linalg.norm(Xs)
        # the old sweep-granular rule is still there and never stops earlier
        _, info2 = lyap_adi_shift_parallel(ops, ms, torch.from_numpy(W.copy()), adi_max_steps=200,
                                           adi_newZ_reltol=tol, width=G, col_parts=parts, stop_rule="sweep")
        assert info2["adi_steps"] >= info["adi_steps"] and info2["adi_steps"] % G == 0


def test_rejects_repeated_shift_in_sweep():
    pr, F, W, tb = _problem(4)
    ops = OracleOps(F.T.tocsr(), pr.M.T.tocsr(), pr.J)
    with pytest.raises(ValueError):
        lyap_adi_shift_parallel(ops, [-1.0, -2.0], torch.from_numpy(W.copy()), width=3)


def test_recycle_depth_restored_also_by_an_exception():
    """the sweep driver raises the contexts' recycling depth for its own sweeps (at least 3, or the
    caller's own if deeper) and restores what the caller had set -- also when the call ends in an exception."""
    pr, F, W, tb = _problem(4)

    class Cx:
        def __init__(self, d):
            self.recycle_depth, self.seen = d, []

        def set_recycle(self, d):
            self.seen.append(d)
            self.recycle_depth = d

    class Ops(OracleOps):
        pass
    for user_depth, during in ((0, 3), (5, 5)):
        ops = Ops(F.T.tocsr(), pr.M.T.tocsr(), pr.J)
        ops.ctx = Cx(user_depth)
        ops.ctxs = [ops.ctx]
        with pytest.raises(ValueError):
            lyap_adi_shift_parallel(ops, [-1.0, -2.0], torch.from_numpy(W.copy()), width=3)
        assert ops.ctx.seen == [during, user_depth] and ops.ctx.recycle_depth == user_depth
        lyap_adi_shift_parallel(ops, [-1.0, -2.0, -4.0], torch.from_numpy(W.copy()), width=3, adi_max_steps=3)
        assert ops.ctx.seen[-2:] == [during, user_depth]


def test_sweep_width_shrinks_when_cauchy_matrix_is_singular():
    """A shift cycle with a repeated shift cannot be swept 4 wide (singular Cauchy matrix):
    the width is halved until every sweep is admissible, the result is the sequential one."""
    pr, F, W, tb = _problem(4)
    ops = OracleOps(F.T.tocsr(), pr.M.T.tocsr(), pr.J)
    ms = [-1.0, -3.0, -1.0, -9.0]
    blocks, info = lyap_adi_shift_parallel(ops, ms, torch.from_numpy(W.copy()), width=4,
                                           adi_max_steps=8, adi_newZ_reltol=0.0)
    assert info["width"] == 2 and info["adi_steps"] == 8
    Zb = torch.cat(blocks, dim=1).numpy()
    Zs = opru.solve_proj_lyap_stein(amat=F, mmat=pr.M, jmat=pr.J, wmat=W,
                                    adi_dict=dict(adi_max_steps=8, adi_newZ_reltol=0.0, ms=ms))
    Ks = opru.get_mTzzTtb(pr.M.T, Zs["zfac"], tb)
    Kb = opru.get_mTzzTtb(pr.M.T, Zb, tb)
    assert np.linalg.norm(Kb - Ks) <= 1e-10 * np.linalg.norm(Ks)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, outdir, width=4, col_parts=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pr, F, W, tb = _problem()
        ms = pb.logshifts(1.0, 500.0, 8)
        ops = OracleOps(F.T.tocsr(), pr.M.T.tocsr(), pr.J)
        blocks, info = lyap_adi_shift_parallel(ops, ms, torch.from_numpy(W.copy()),
                                               adi_max_steps=200, adi_newZ_reltol=1e-9,
                                               width=width, col_parts=col_parts)
        Zb = torch.cat(blocks, dim=1).numpy()
        K = opru.get_mTzzTtb(pr.M.T, Zb, tb)
        # every rank must hold the same replicated result
        Kt = torch.from_numpy(K.copy())
        ref = Kt.clone()
        dist.broadcast(ref, src=0)
        same = bool(torch.allclose(ref, Kt, rtol=0, atol=0))
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), K=K, steps=info["adi_steps"],
                 solved=len(ops.lus), same=same, parts=info["col_parts"],
                 owners=np.array(info["owners"] if info["owners"] is not None else [-1]),
                 mine=np.array(sorted(ops.lus)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo(tmp_path):
    """world_size 2, width 4: each rank solves 2 shifts per sweep; K matches the
    sequential oracle within the parity bar (1e-6 rel. Frobenius)."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert bool(r0["same"]) and bool(r1["same"])
    assert np.array_equal(r0["K"], r1["K"])
    # each rank only ever factorised its OWN shifts: the fixed owners of the library's dealing
    # (ricadi_host_deal -- the same table the sharded sweeps inside libricadi_hip.so use)
    from optconpy_amd import _lib
    ms8 = pb.logshifts(1.0, 500.0, 8)
    own = _lib.host_deal(ms8, 2)
    assert np.array_equal(r0["owners"], own) and np.array_equal(r1["owners"], own)
    assert np.allclose(r0["mine"], sorted(p for p, o in zip(ms8, own) if o == 0))
    assert np.allclose(r1["mine"], sorted(p for p, o in zip(ms8, own) if o == 1))
    assert int(r0["solved"]) + int(r1["solved"]) == 8
    pr, F, W, tb = _problem()
    ms = pb.logshifts(1.0, 500.0, 8)
    d = dict(adi_max_steps=200, adi_newZ_reltol=1e-9, ms=ms)
    Zs = opru.solve_proj_lyap_stein(amat=F, mmat=pr.M, jmat=pr.J, wmat=W, adi_dict=d)["zfac"]
    Ks = opru.get_mTzzTtb(pr.M.T, Zs, tb)
    assert np.linalg.norm(r0["K"] - Ks) <= 1e-6 * np.linalg.norm(Ks)


@pytest.mark.parametrize("width,col_parts", [(4, 2), (1, 0)])
def test_two_ranks_gloo_column_split(tmp_path, width, col_parts):
    """world_size 2 with column parts: (4 shifts x 2 parts per sweep, 4 items per rank) and
    (1 shift per sweep -> automatically 2 parts, so that the second rank has work).  All-gather
    into the rank-major buffer, recombination with permuted coefficients; K as the oracle's."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), width, col_parts), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert bool(r0["same"]) and bool(r1["same"]) and np.array_equal(r0["K"], r1["K"])
    assert int(r0["parts"]) == 2
    pr, F, W, tb = _problem()
    ms = pb.logshifts(1.0, 500.0, 8)
    d = dict(adi_max_steps=200, adi_newZ_reltol=1e-9, ms=ms)
    Zs = opru.solve_proj_lyap_stein(amat=F, mmat=pr.M, jmat=pr.J, wmat=W, adi_dict=d)["zfac"]
    Ks = opru.get_mTzzTtb(pr.M.T, Zs, tb)
    assert np.linalg.norm(r0["K"] - Ks) <= 1e-6 * np.linalg.norm(Ks)


def test_truncated_sweep_keeps_residual_factor_consistent():
    """A sweep that stops before its last block (kept < G) must leave W = the residual factor of the
    TRUNCATED Z: the reported residual then equals the factored residual of the returned factor
    (W used to be advanced with the Cauchy data of all G solves)."""
    pr, F, W, tb = _problem()
    ms = pb.logshifts(1.0, 500.0, 8)
    ops = OracleOps(F.T.tocsr(), pr.M.T.tocsr(), pr.J)
    W0 = olau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=W, transposedprj=True)
    # a tolerance the iteration meets in the middle of one of its first two sweeps: no block-norm history
    # yet, so the sweep is NOT cut at the predicted stopping step and blocks are really dropped
    tol = None
    for cand in (0.5, 0.3, 0.2, 0.1, 0.05, 0.02, 0.01):
        blocks, info = lyap_adi_shift_parallel(OracleOps(F.T.tocsr(), pr.M.T.tocsr(), pr.J), ms,
                                               torch.from_numpy(W0.copy()), adi_max_steps=40,
                                               adi_newZ_reltol=cand, width=8)
        if info["adi_steps"] % 8 != 0 and info["sweeps"] <= 2:
            tol = cand
            break
    assert tol is not None, "no tolerance found that stops inside a sweep"
    Zb = torch.cat(blocks, dim=1).numpy()
    res2 = opru.comp_proj_lyap_res_norm(Zb, F, pr.M, W0, pr.J)
    assert np.isclose(info["res_fro"], np.sqrt(res2), rtol=1e-6), (info["res_fro"], np.sqrt(res2))
