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

    def _batch_on(self, ctx, pairs):
        """One batched solve of the (shift, panel) pairs on ``ctx``; returns the n x m' solution
        panels (device views), the iteration counts and the true residuals."""
        m = pairs[0][1].shape[1]
        ps = [float(p) for p, _ in pairs]
        shared = all(W is pairs[0][1] for _, W in pairs)
        if shared:
            R, stride = pairs[0][1], 0
        else:
            R = torch.stack([W for _, W in pairs], dim=0).contiguous()
            stride = R.shape[1] * m
        X = self.empty(len(ps), ctx.n, m)
        torch.cuda.current_stream().synchronize()           # R is ready before ricadi's stream reads it
        its, rr = ctx.shift_solve_batch_dev(ps, [1.0] * len(ps), R.data_ptr(), stride, m, X.data_ptr(),
                                            strict=False)
        ctx.synchronize()
        return X, its, rr

    def solve_items(self, pairs, U_out):
        """Solve the work items ``(shift, NV x m' panel)`` of this rank and write the velocity
        rows of solution k into ``U_out[k]``.  One batched lockstep solve (chunks of 16); with
        extra contexts the list is cut into one batch per context, run concurrently."""
        import time
        if not pairs:
            return
        self._sync_in()
        t0 = time.perf_counter()
        nctx = min(len(self.ctxs), len(pairs))
        nv = self.ctx.nv
        if nctx <= 1:
            for c0 in range(0, len(pairs), self.MAX_BATCH):
                chunk = pairs[c0:c0 + self.MAX_BATCH]
                X, its, rr = self._batch_on(self.ctx, chunk)
                for k in range(len(chunk)):
                    U_out[c0 + k].copy_(X[k, :nv])
                self.gmres_iters += int(sum(its))
                self._account(rr)
        else:
            from concurrent.futures import ThreadPoolExecutor
            if self._pool is None:
                self._pool = ThreadPoolExecutor(max_workers=len(self.ctxs))
            bounds = [round(i * len(pairs) / nctx) for i in range(nctx + 1)]
            import os
            if nctx == 2 and os.environ.get("RICADI_SPLIT"):      # developer probe: slow / fast split point
                bounds = [0, min(len(pairs) - 1, max(1, int(os.environ["RICADI_SPLIT"]))), len(pairs)]

            def lane(k):
                return self._batch_on(self.ctxs[k], pairs[bounds[k]:bounds[k + 1]])

            for k, (X, its, rr) in enumerate(self._pool.map(lane, range(nctx))):
                for j in range(bounds[k + 1] - bounds[k]):
                    U_out[bounds[k] + j].copy_(X[j, :nv])
                self.gmres_iters += int(sum(its))
                self._account(rr)
        torch.cuda.current_stream().synchronize()
        self.shift_solves += len(pairs)
        self.t_solve += time.perf_counter() - t0

    def solve_many(self, ps, W):
        """Solve the same panel ``W`` against several shifts; list of NV x m tensors."""
        if not len(ps):
            return []
        U = self.empty(len(ps), W.shape[0], W.shape[1])
        self.solve_items([(p, W) for p in ps], U)
        return [U[k] for k in range(len(ps))]

    def recombine_slots(self, U_all, coefz, coefw, W):
        """One device call for the Cauchy recombination of a sweep from the gathered buffer
        ``U_all`` (nslot x NV x m'): returns the NV x (G*m') block and its squared Frobenius
        norm, and updates ``W += E sum_i coefw[i] U_i`` in place."""
        nslot, nv, m = U_all.shape
        G = np.asarray(coefz).shape[1]
        Zb = self.empty(nv, G * m)
        self._sync_in()
        n2, bn = self.ctx.sweep_recombine_slots_dev(nslot, G, U_all.data_ptr(), m, coefz, coefw,
                                                    Zb.data_ptr(), W.data_ptr())
        return Zb, n2, bn

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
        """Recombination with the solutions in sweep order (``U_all``: G x NV x m)."""
        Zb, n2, _ = self.recombine_slots(U_all, np.asarray(rinv), np.asarray(cinv1), W)
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


def plan_items(G, world, col_parts=0):
    """Column parts per shift.  The work items of a sweep are the pairs (shift g, column
    part q): the m columns of the residual factor are independent for a fixed shift
    ("alternative axis"), so a sweep offers ``G * parts`` items.
    ``col_parts > 0`` fixes the number; automatic: 1 while every rank gets a shift of its
    own, else the smallest count that gives every rank an item (more ranks than shifts).
    A part narrower than 16 columns runs the 16-lane-row kernels with idle lanes, so parts
    are not used to shorten the per-rank batch when the shifts already cover the ranks."""
    if col_parts and col_parts > 0:
        return int(col_parts)
    return max(1, -(-world // G))


def item_layout(G, parts, world, owners=None):
    """Dealing of the ``G * parts`` items of a sweep.  Round robin (``owners=None``): item
    ``i = g * parts + q`` goes to rank ``i % world``, local slot ``i // world``.  With ``owners`` (one rank
    per shift of the sweep, ``parts == 1``): the FIXED owners of ``_lib.host_deal`` -- the dealing the library's
    own sharded sweeps use (``ricadi_set_exchange``): slow small-|p| shifts alone, fast ones stacked, and a shift
    never changes its rank, so that its per-shift setup and recycled solutions stay where they are.  In the
    rank-major buffer an all-gather fills an item sits at ``rank * per_rank + slot``."""
    nitems = G * parts
    items = []
    if owners is not None:
        if parts != 1 or len(owners) != G:
            raise ValueError("fixed owners need one rank per shift and a single column part")
        cnt = [0] * world
        for g in range(G):
            r = int(owners[g])
            items.append(dict(g=g, q=0, rank=r, slot=cnt[r]))
            cnt[r] += 1
        per_rank = max(1, max(cnt))
    else:
        per_rank = -(-nitems // world)
        for i in range(nitems):
            items.append(dict(g=i // parts, q=i % parts, rank=i % world, slot=i // world))
    for it in items:
        it["pos"] = it["rank"] * per_rank + it["slot"]
    return items, per_rank


def lyap_adi_shift_parallel(ops, ms, W, **kw):
    """Shift-parallel LR-ADI (:func:`_lyap_adi_sweeps`, which documents the arguments) with the library's recycled
    initial guesses switched on for the sweeps of this call: they solve nearly the same right-hand-side space
    again and again, so every batched solve starts from the least-squares combination of its last solved panels
    (``ricadi_set_recycle``, depth 3 or the caller's own if deeper; the C++ drivers do the same for their sweeps).
    The contexts' depths are restored on the way out, also by an exception."""
    ctxs = [cx for cx in getattr(ops, "ctxs", [getattr(ops, "ctx", None)]) if hasattr(cx, "set_recycle")]
    prev = [int(getattr(cx, "recycle_depth", 0)) for cx in ctxs]
    try:
        for cx, d in zip(ctxs, prev):
            cx.set_recycle(max(3, d))
        return _lyap_adi_sweeps(ops, ms, W, **kw)
    finally:
        for cx, d in zip(ctxs, prev):
            cx.set_recycle(d)


def _lyap_adi_sweeps(ops, ms, W, adi_max_steps=200, adi_newZ_reltol=1e-8,
                     group=None, width=None, max_width=8, verbose=False, col_parts=0,
                     stop_rule="step"):
    """Shift-parallel LR-ADI; returns ``(Z_blocks, info)``.

    ``W`` is the (already projected) NV x m residual factor as a tensor on the
    ops' device, replicated on every rank.  Each sweep handles up to ``G`` distinct
    shifts, ``G = width`` or ``min(world_size, len(ms), max_width)`` (Cauchy
    conditioning limits ``G``: SURVEY.md F8).  The sweep's work items -- (shift,
    column part) pairs, :func:`plan_items` / :func:`item_layout` -- are dealt to the
    ranks; a rank solves all its items in ONE batched solve, the solutions are
    all-gathered into one preallocated rank-major buffer that the recombination
    reads in place (the Cauchy coefficients are permuted, not the data).

    Stopping (``stop_rule="step"``): with ``C = R^T R``, column block ``j`` of ``U R^-1`` lies
    in ``span{U_1..U_j}`` -- it is the block the step-by-step iteration appends at step ``j``
    -- so the reference's rule, relative norm of the new block below ``adi_newZ_reltol``
    (``optcont_main.py:123-124``), is applied block by block: the iteration ends after the
    same step as the sequential one and the blocks behind it are dropped.  The block norms
    of the last two passes over the shift cycle predict the stopping step, and the sweep
    that would contain it is cut there.  ``stop_rule="sweep"``: mean block norm per sweep.
    Rank 0 takes the decisions for everybody.
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
    nv, m = W.shape
    parts = plan_items(G, world, col_parts)
    if m % parts:
        raise ValueError("panel width {0} is not divisible into {1} column parts".format(m, parts))
    mp_ = m // parts
    # residual factor kept as `parts` contiguous NV x m' panels (the right-hand sides of the
    # items and the operands of the recombination)
    Wq = [W[:, q * mp_:(q + 1) * mp_].contiguous().clone() for q in range(parts)]
    buffers = {}                      # per (first step, sweep width): (items, per_rank, U_loc, U_all)
    # fixed owner of every shift of the list (the library's dealing) where the shifts cover the ranks
    fixed = None
    if world > 1 and parts == 1 and len(set(float(p) for p in ms)) == ns:
        fixed = [int(o) for o in _lib.host_deal(ms, world)]

    def layout(first, g):
        key = (first % ns, g) if fixed is not None else g
        if key not in buffers:
            own = None if fixed is None else [fixed[(first + j) % ns] for j in range(g)]
            items, per_rank = item_layout(g, parts, world, own)
            U_loc = W.new_zeros((per_rank, nv, mp_))          # padding slots stay zero
            U_all = W.new_zeros((world * per_rank, nv, mp_)) if world > 1 else U_loc
            buffers[key] = (items, per_rank, U_loc, U_all)
        return buffers[key]

    step_rule = stop_rule == "step" and adi_newZ_reltol > 0.0
    rel_h1, rel_h2 = np.zeros(ns), np.zeros(ns)
    blocks = []
    znorm2 = 0.0
    steps = 0
    rel = float("inf")
    nsweeps = 0
    items_solved = 0
    while steps < adi_max_steps:
        # width of this sweep: cut at the predicted stopping step (see the docstring)
        g_now = min(G, adi_max_steps - steps)
        if step_rule:
            for g in range(g_now):
                pos = (steps + g) % ns
                if rel_h1[pos] > 0.0 and rel_h2[pos] > rel_h1[pos] and \
                        rel_h1[pos] * (rel_h1[pos] / rel_h2[pos]) < adi_newZ_reltol:
                    g_now = g + 1
                    break
        if world > 1:
            gw = W.new_tensor([float(g_now)])
            dist.broadcast(gw, src=0, group=group)
            g_now = int(round(gw.item()))
        ps = [float(ms[(steps + g) % ns]) for g in range(g_now)]
        if len(set(ps)) != g_now:
            raise ValueError("shifts within one sweep must be distinct: {0}".format(ps))
        items, per_rank, U_loc, U_all = layout(steps, g_now)
        if world > 1 and len([it for it in items if it["rank"] == rank]) < per_rank:
            U_loc.zero_()                 # padding slots of THIS sweep travel as zeros
        mine = [it for it in items if it["rank"] == rank]
        nslot = world * per_rank
        if hasattr(ops, "solve_items"):
            ops.solve_items([(ps[it["g"]], Wq[it["q"]]) for it in mine], U_loc)
        else:
            for it in mine:
                U_loc[it["slot"]].copy_(ops.solve(ps[it["g"]], Wq[it["q"]]))
        items_solved += len(mine)
        if world > 1:
            dist.all_gather_into_tensor(U_all.view(-1), U_loc.view(-1), group=group)
        rinv, cinv1 = _lib.host_cauchy(ps)
        bn2 = np.zeros(g_now)             # squared norm of every (sequential) block of the sweep
        zparts = []
        for q in range(parts):
            # coefficients in BUFFER order: slot `pos` carries shift g (other parts / padding: 0)
            coefz = np.zeros((nslot, g_now))
            coefw = np.zeros(nslot)
            for it in items:
                if it["q"] == q:
                    coefz[it["pos"], :] = rinv[it["g"], :]
                    coefw[it["pos"]] = cinv1[it["g"]]
            if hasattr(ops, "recombine_slots"):
                Zq, _, bq = ops.recombine_slots(U_all, coefz, coefw, Wq[q])
                zparts.append([Zq[:, j * mp_:(j + 1) * mp_] for j in range(g_now)])
                bn2 += np.asarray(bq)
            else:
                zj = []
                for j in range(g_now):
                    Zj = ops.lincomb(coefz[:, j], U_all)
                    bn2[j] += ops.fro2(Zj)
                    zj.append(Zj)
                zparts.append(zj)
                T = ops.lincomb(coefw, U_all)
                ops.apply_E(1.0, T, Wq[q])
        # stopping decision (rank 0's numbers: the norms come from kernels with atomic
        # accumulation, ranks may differ in the last bits and must not disagree)
        kept, stop = g_now, False
        if step_rule:
            z2 = znorm2
            for j in range(g_now):
                z2 += bn2[j]
                rel = float(np.sqrt(bn2[j] / z2)) if z2 > 0 else 0.0
                pos = (steps + j) % ns
                rel_h2[pos], rel_h1[pos] = rel_h1[pos], rel
                if rel < adi_newZ_reltol:
                    kept, stop = j + 1, True
                    break
        else:
            n2 = float(bn2.sum())
            rel = float(np.sqrt(n2 / g_now / (znorm2 + n2))) if znorm2 + n2 > 0 else 0.0
            stop = rel < adi_newZ_reltol
        if world > 1:
            dec = W.new_tensor([float(kept), 1.0 if stop else 0.0, rel])
            dist.broadcast(dec, src=0, group=group)
            kept, stop, rel = int(round(dec[0].item())), bool(dec[1].item() > 0.5), float(dec[2].item())
        if kept < g_now:
            # W was advanced with C^-1 1 of all g_now shifts; the factor keeps only the first `kept` blocks.
            # Every U_g was solved against the same W, so the kept solutions ARE the sweep of the first
            # `kept` shifts, whose Cauchy data differ only in C^-1 1: correct W by the difference, so that
            # it stays the residual factor of the truncated Z (and `res_fro` / `resfac` its residual).
            _, cinv_k = _lib.host_cauchy(ps[:kept])
            for q in range(parts):
                delta = np.zeros(nslot)
                for it in items:
                    if it["q"] == q:
                        delta[it["pos"]] = (cinv_k[it["g"]] if it["g"] < kept else 0.0) - cinv1[it["g"]]
                ops.apply_E(1.0, ops.lincomb(delta, U_all), Wq[q])
        znorm2 += float(bn2[:kept].sum())
        for j in range(kept):
            for q in range(parts):
                blocks.append(zparts[q][j])
        steps += kept
        nsweeps += 1
        if verbose and rank == 0:
            print("sweep {0:3d}: {1} shifts, kept {2}, rel new Z {3:9.3e}".format(nsweeps, g_now, kept, rel))
        if stop:
            break
    Wend = Wq[0] if parts == 1 else torch.cat(Wq, dim=1).contiguous()
    info = dict(adi_steps=steps, sweeps=nsweeps, width=G, col_parts=parts, adi_rel_newZ=rel, owners=fixed,
                res_fro=ops.gram_fro(Wend), resfac=Wend,
                gmres_nonconverged=int(getattr(ops, "nonconverged", 0)),
                gmres_worst_relres=float(getattr(ops, "worst_relres", 0.0)),
                shift_solves=items_solved / float(parts))
    if world > 1:
        # a rank only sees its own solves: the counts are summed, the worst residual maximised
        t = W.new_tensor([info["gmres_nonconverged"], info["shift_solves"]], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        w = W.new_tensor([info["gmres_worst_relres"]], dtype=torch.float64)
        dist.all_reduce(w, op=dist.ReduceOp.MAX, group=group)
        info["gmres_nonconverged"], info["shift_solves"] = int(t[0].item()), float(t[1].item())
        info["gmres_worst_relres"] = float(w.item())
    if info["gmres_nonconverged"] and rank == 0:
        _lib._warn_nonconverged(info)
    return blocks, info
