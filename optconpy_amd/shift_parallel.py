"""Shift-parallel low-rank ADI: one ADI shift per GPU, one all-gather per sweep.

The textbook LR-ADI recurrence is sequential in the shifts.  Its Cauchy form
 is not: in a *sweep* of ``G`` distinct real
shifts every rank ``g`` solves

    S(p_g) [U_g; L] = [W_s; 0]

against the **same** residual factor ``W_s`` (operator replicated, no data-path
dependency), the ``U_g`` are all-gathered (RCCL over xGMI on the GPUs, gloo in
the CPU tests), and every rank recombines them redundantly with the ``G x G``
Cauchy matrix ``C_ij = -1/(p_i + p_j) = R^T R``:

    Z-block  = U (R^-1 (x) I_m),        W_{s+1} = W_s + E U ((C^-1 1) (x) I_m)

which reproduces ``G`` sequential ADI steps exactly (up to a rotation of the
block's columns; ``Z Z^T`` and hence the gain ``K`` are identical).

The reference has nothing distributed; this is new
design.  The local operations are behind a small ``ops`` object so the same
orchestration runs on ``torch.cuda`` tensors + ``libricadi_hip.so``
(:class:`HipOps`) and, in the CPU test-suite, on CPU tensors with injected
solves.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


class HipOps:
    """Local panel operations on device tensors through the C-ABI.

    ``extra_ctxs``: further contexts holding the SAME operator on the same GPU.
    Each context owns a HIP stream; the shifts a rank has to solve in one sweep
    are then solved concurrently, one host thread and one stream per shift
    (``solve_many``).  At n ~ 3e4 a single shift-solve is a chain of short,
    latency-bound kernels that leaves most of the chip idle, so concurrent
    streams raise the throughput almost linearly for a few streams.
    """

    def __init__(self, ctx, extra_ctxs=()):
        self.ctx = ctx
        self.ctxs = [ctx] + list(extra_ctxs)
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.gmres_iters = 0
        self.shift_solves = 0
        self.nonconverged = 0       # shift-solves that stopped at gmres_maxit above the tolerance
        self.worst_relres = 0.0     # worst true relative residual seen (all solves)
        self.t_solve = 0.0          # wall seconds inside solve_many / solve (synchronised)
        self._pool = None

    def _account(self, relres):
        """Book-keeping of the true FP64 residuals a solve returned (one row per shift):
        a solve that missed ``gmres_tol`` is folded into Z like any other -- as in the
        C++ driver -- but it is counted and reported (``info`` of the sweep loop)."""
        rr = np.atleast_2d(np.asarray(relres, dtype=float))
        tol = float(getattr(getattr(self.ctx, "_opts", None), "gmres_tol", 1e-10))
        worst = rr.max(axis=1)
        self.nonconverged += int(np.sum(~(worst <= tol * 1.0000001)))
        self.worst_relres = max(self.worst_relres, float(worst.max()))

    def set_lowrank(self, U=None, V=None):
        for c in self.ctxs:
            c.set_lowrank(U, V)

    def clear_cache(self):
        for c in self.ctxs:
            c.clear_cache()

    def _solve_on(self, ctx, p, W):
        m = W.shape[1]
        X = self.empty(ctx.n, m)
        its, rr = ctx.shift_solve_dev(float(p), 1.0, W.data_ptr(), m, X.data_ptr(), strict=False)
        ctx.synchronize()
        self._account(rr)
        return X[:ctx.nv].contiguous(), its

    MAX_BATCH = 16

    def solve_many(self, ps, W):
        """Solve the same panel ``W`` against several shifts.  Default: ONE batched
        solve (``ricadi_shift_solve_batch_dev``) -- all shifts advance in lockstep
        inside one launch sequence, grid.z = shifts still iterating.  With extra
        contexts (``extra_ctxs``) the list is cut into one batch per context and the
        batches run concurrently on their streams."""
        import time
        nctx = len(self.ctxs)
        if len(ps) <= 1:
            return [self.solve(p, W) for p in ps]
        if nctx == 1:
            self._sync_in()
            t0 = time.perf_counter()
            ctx = self.ctx
            m = W.shape[1]
            out = []
            for c0 in range(0, len(ps), self.MAX_BATCH):
                chunk = [float(p) for p in ps[c0:c0 + self.MAX_BATCH]]
                X = self.empty(len(chunk), ctx.n, m)
                its, rr = ctx.shift_solve_batch_dev(chunk, [1.0] * len(chunk), W.data_ptr(), 0, m,
                                                    X.data_ptr(), strict=False)
                ctx.synchronize()
                self._account(rr)
                out.extend(X[g, :ctx.nv].contiguous() for g in range(len(chunk)))
                self.gmres_iters += int(sum(its))
                self.shift_solves += len(chunk)
            self.t_solve += time.perf_counter() - t0
            return out
        # several contexts: the shift list is cut into len(ctxs) contiguous chunks, each
        # chunk is ONE batched solve on its context's stream, driven by its own host thread
        # (ctypes releases the GIL).  While one batch is down to a few active groups --
        # short, latency-bound kernels -- the other batch's kernels fill the chip.
        from concurrent.futures import ThreadPoolExecutor
        if self._pool is None:
            self._pool = ThreadPoolExecutor(max_workers=nctx)
        self._sync_in()
        t0 = time.perf_counter()
        m = W.shape[1]
        nchunk = min(nctx, len(ps))
        bounds = [round(i * len(ps) / nchunk) for i in range(nchunk + 1)]

        def lane(k):
            ctx = self.ctxs[k]
            chunk = [float(p) for p in ps[bounds[k]:bounds[k + 1]]]
            X = self.empty(len(chunk), ctx.n, m)
            its, rr = ctx.shift_solve_batch_dev(chunk, [1.0] * len(chunk), W.data_ptr(), 0, m,
                                                X.data_ptr(), strict=False)
            ctx.synchronize()
            return [X[g, :ctx.nv].contiguous() for g in range(len(chunk))], int(sum(its)), rr

        out = []
        for Us, its, rr in self._pool.map(lane, range(nchunk)):
            out.extend(Us)
            self.gmres_iters += its
            self._account(rr)
        self.shift_solves += len(ps)
        self.t_solve += time.perf_counter() - t0
        return out

    def to_panel(self, W):
        return torch.as_tensor(np.ascontiguousarray(W), dtype=torch.float64).to(self.device)

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float64, device=self.device)

    def _sync_in(self):
        # torch's stream -> ricadi's stream hand-off (rare: a few times per sweep)
        torch.cuda.current_stream().synchronize()

    def solve(self, p, W):
        """First NV rows of ``S(p,1)^-1 [W; 0]`` as a new NV x m tensor."""
        self._sync_in()
        U, its = self._solve_on(self.ctx, p, W)
        self.gmres_iters += its
        self.shift_solves += 1
        return U

    def recombine(self, U_all, rinv, cinv1, W):
        """One device call for the Cauchy recombination of a sweep: returns the
        NV x (G*m) block ``U (R^-1 (x) I)`` and its squared Frobenius norm, and
        updates ``W += E U ((C^-1 1) (x) I)`` in place."""
        G, nv, m = U_all.shape
        Zb = self.empty(nv, G * m)
        self._sync_in()
        n2 = self.ctx.sweep_recombine_dev(G, U_all.data_ptr(), m, rinv, cinv1, Zb.data_ptr(),
                                          W.data_ptr())
        return Zb, n2

    def lincomb(self, coef, U_all):
        """``sum_i coef[i] * U_all[i]``; ``U_all`` is G x NV x m, contiguous."""
        G, nv, m = U_all.shape
        out = self.empty(nv, m)
        self._sync_in()
        self.ctx.lincomb_dev(nv, m, coef, U_all.data_ptr(), nv * m, out.data_ptr())
        self.ctx.synchronize()
        return out

    def apply_E(self, coef, V, W):
        """``W += coef * E * V`` in place."""
        self._sync_in()
        self.ctx.apply_e_dev(float(coef), V.data_ptr(), V.shape[1], W.data_ptr())
        self.ctx.synchronize()

    def gain(self, coef, Z, B):
        """``coef * E * (Z * (Z^T B))`` for a contiguous NV x c factor ``Z``."""
        K = self.empty(B.shape[0], B.shape[1])
        self._sync_in()
        self.ctx.gain_dev(float(coef), Z.data_ptr(), Z.shape[1], Z.shape[1], B.data_ptr(),
                          B.shape[1], K.data_ptr())
        return K

    def fro2(self, T):
        self._sync_in()
        _, t = self.ctx.panel_norms_dev(T.data_ptr(), T.shape[0], T.shape[1])
        return t

    def gram_fro(self, T):
        self._sync_in()
        g, _ = self.ctx.panel_norms_dev(T.data_ptr(), T.shape[0], T.shape[1])
        return g


def sweep_shifts(ms, sweep, G):
    """Shifts of sweep ``sweep``: ADI steps ``sweep*G+1 .. (sweep+1)*G`` of the cycle."""
    ns = len(ms)
    return [float(ms[(sweep * G + g) % ns]) for g in range(G)]


def lyap_adi_shift_parallel(ops, ms, W, adi_max_steps=200, adi_newZ_reltol=1e-8,
                            group=None, width=None, max_width=8, verbose=False):
    """Shift-parallel LR-ADI; returns ``(Z_blocks, info)``.

    ``W`` is the (already projected) NV x m residual factor as a tensor on the
    ops' device, replicated on every rank.  Each sweep handles ``G`` distinct
    shifts, ``G = width`` or ``min(world_size, len(ms), max_width)`` (Cauchy
    conditioning limits ``G``: SURVEY.md F8); shift ``g`` of the sweep is
    solved by rank ``g % world_size``, so ``G`` may exceed the number of ranks.
    Stops after the sweep in which the mean new-block norm falls below
    ``adi_newZ_reltol`` (the sequential rule of ``optcont_main.py:123-124`` at
    sweep granularity) or after ``adi_max_steps`` steps.
    """
    if dist.is_available() and dist.is_initialized():
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
    else:
        world, rank = 1, 0
    ns = len(ms)
    G = int(width) if width else max(1, min(world, ns, max_width))
    if G > ns:
        raise ValueError("sweep width {0} exceeds the number of shifts {1}".format(G, ns))
    # A sweep needs a numerically positive definite Cauchy matrix of its shifts: halve the
    # width until every sweep of the shift cycle has one (deterministic, the same on all ranks)
    while G > 1:
        try:
            for sw in range(ns):
                _lib.host_cauchy(sweep_shifts(ms, sw, G))
            break
        except (RuntimeError, ValueError):
            G = max(1, G // 2)
    per_rank = (G + world - 1) // world
    nv, m = W.shape
    W = W.clone()
    blocks = []
    znorm2 = 0.0
    steps = 0
    rel = float("inf")
    nsweeps = 0
    while steps + G <= adi_max_steps or steps == 0:
        ps = sweep_shifts(ms, nsweeps, G)
        if len(set(ps)) != G:
            raise ValueError("shifts within one sweep must be distinct: {0}".format(ps))
        mine = [g for g in range(G) if g % world == rank]
        if hasattr(ops, "solve_many"):
            local = ops.solve_many([ps[g] for g in mine], W)
        else:
            local = [ops.solve(ps[g], W) for g in mine]
        while len(local) < per_rank:
            local.append(torch.zeros_like(W))
        U_loc = torch.stack(local, dim=0).contiguous()
        if world > 1:
            gathered = [torch.empty_like(U_loc) for _ in range(world)]
            dist.all_gather(gathered, U_loc, group=group)
            U_all = torch.stack([gathered[g % world][g // world] for g in range(G)],
                                dim=0).contiguous()
        else:
            U_all = U_loc[:G].contiguous()
        rinv, cinv1 = _lib.host_cauchy(ps)
        if hasattr(ops, "recombine"):
            Zb, n2 = ops.recombine(U_all, rinv, cinv1, W)
            blocks.append(Zb)
        else:
            n2 = 0.0
            for j in range(G):
                Zj = ops.lincomb(rinv[:, j], U_all)
                n2 += ops.fro2(Zj)
                blocks.append(Zj)
            T = ops.lincomb(cinv1, U_all)
            ops.apply_E(1.0, T, W)
        znorm2 += n2
        steps += G
        nsweeps += 1
        rel = float(np.sqrt(n2 / G / znorm2)) if znorm2 > 0 else 0.0
        if verbose and rank == 0:
            print("sweep {0:3d}: shifts {1} rel new Z {2:9.3e}".format(nsweeps, ps, rel))
        stop = rel < adi_newZ_reltol
        if world > 1:
            # The norms above come from kernels with atomic accumulation: ranks may
            # differ in the last bits and must not disagree on leaving the loop
            # (the next all-gather would hang).  Rank 0 decides for everybody.
            flag = W.new_tensor([1.0 if stop else 0.0])
            dist.broadcast(flag, src=0, group=group)
            stop = bool(flag.item() > 0.5)
        if stop:
            break
    info = dict(adi_steps=steps, sweeps=nsweeps, width=G, adi_rel_newZ=rel,
                res_fro=ops.gram_fro(W), resfac=W,
                gmres_nonconverged=int(getattr(ops, "nonconverged", 0)),
                gmres_worst_relres=float(getattr(ops, "worst_relres", 0.0)),
                shift_solves=int(getattr(ops, "shift_solves", 0)))
    if world > 1:
        # a rank only sees its own solves: the counts are summed, the worst residual maximised
        t = W.new_tensor([info["gmres_nonconverged"], info["shift_solves"]])
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        w = W.new_tensor([info["gmres_worst_relres"]])
        dist.all_reduce(w, op=dist.ReduceOp.MAX, group=group)
        info["gmres_nonconverged"], info["shift_solves"] = int(t[0].item()), int(t[1].item())
        info["gmres_worst_relres"] = float(w.item())
    if info["gmres_nonconverged"] and rank == 0:
        _lib._warn_nonconverged(info)
    return blocks, info
