#!/usr/bin/env python
"""Benchmark of the low-rank Newton-ADI hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): ADI shift-solves per second, next to the wall-clock to
the feedback gain K.  One *step* = one Newton step of the projected Riccati
solve on BASELINE config 2 (driven cavity, N=58 -> n = 29 930, nu = 0.05,
16 log-spaced ADI shifts, right-hand-side panel m = NY' + NU = 16): a complete
low-rank ADI solve of the closed-loop Lyapunov equation with the reference's
default stopping rule (adi_newZ_reltol = 1e-8, optcont_main.py:124) followed by
the gain K = -E Z Z^T B.  One *unit* = one shift-solve, i.e. one saddle-point
solve S(p) [V;L] = [R;0] with an NV x 16 panel to relative residual 1e-11.

Every N runs the same problem with the shift-parallel Cauchy sweeps
(optconpy_amd/shift_parallel.py): sweeps of 16 distinct shifts (the whole shift
cycle) solved independently against the same residual factor, recombined with
the 16 x 16 Cauchy matrix -- identical to 16 sequential ADI steps.  Shift g of a
sweep is solved by rank g % N; a rank solves its 16/N shifts in ONE batched GMRES
(ricadi_shift_solve_batch_dev: all shifts advance in lockstep, every kernel of
the iteration is launched once with grid.z = shifts still iterating), since one
shift-solve at this size is a chain of short kernels that leaves most of the
chip idle.  N > 1: one process per GPU (torch.distributed, RCCL), one
all-gather per sweep -> total work fixed, "scaling": "strong".
`--sequential` times the single-panel device-resident C++ ADI instead;
`--streams k` cuts a rank's shifts into k batches that run concurrently on k HIP streams.

The JSON line also carries the SpMM roofline figures (kernel time from HIP
events on the library's stream) and the CPU baseline (oracle = scipy SuperLU on
the same matrices, bounded sample, rank 0 / N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_inputs(N, nu, nshifts):
    """cfg2 inputs, prepared with the product's own host-side mirror (GPU solves)."""
    from optconpy_amd import lin_alg_utils as lau, problems as pb
    pr = pb.ricc_problem(N, nu, NU=4, NY=4, alphau=1e-2)
    mct = lau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=pr.mc_mat.T, transposedprj=True)
    tb = lau.apply_invsqrt_fromright(pr.rmat, pr.b_mat, output="dense")
    trct = lau.apply_invsqrt_fromright(pr.y_masmat, mct, output="dense")
    ms = pb.logshifts(1.0, 3e3, nshifts)
    return pr, tb, trct, ms


def spmm_roofline(ctx, nnz_s, n, m, shifts, reps=200):
    """K1 roofline on the launch the hot path issues: ONE batched tile-SpMM launch over
    the G = len(shifts) panels of a sweep (grid.z = G).  Algorithmic bytes per unit
    (12 nnz + 4 (n+1) + 16 n m) x G units per launch, over the launch's
    average duration measured with HIP events on the ctx stream.  The single-panel launch
    (sequential path) is reported next to it."""
    import torch
    G = len(shifts)
    x = torch.randn(G, n, m, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    al, be = [float(p) for p in shifts], [1.0] * G
    ctx.time_spmm_batch_dev(al, be, x.data_ptr(), m, y.data_ptr(), 20)          # warm-up
    # best of three trials of `reps` launches each (clock dips of a freshly loaded box)
    ms = min(ctx.time_spmm_batch_dev(al, be, x.data_ptr(), m, y.data_ptr(), reps) for _ in range(3))
    ms1 = min(ctx.time_spmm_dev(al[0], 1.0, x.data_ptr(), m, y.data_ptr(), reps) for _ in range(3))
    unit = 12.0 * nnz_s + 4.0 * (n + 1) + 16.0 * n * m
    nbytes = unit * G
    gbs = nbytes / (ms * 1e-3) / 1e9
    # HBM bytes per launch from the committed PMC passes of this kernel and shape
    # (profiles/r01_spmm_traffic.json); None for shapes that were not profiled
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_spmm_traffic.json")) as f:
            traffic = json.load(f).get("%dx%dx%d" % (G, n, m), {}).get("hbm_bytes")
    except (OSError, ValueError):
        pass
    return dict(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(gbs / HBM_PEAK_GBS, 4), traffic=traffic,
                kernel="ricadi::spmm_blocked_kernel", us_per_launch=round(ms * 1e3, 2),
                units_per_launch=G, algorithmic_bytes=int(nbytes),
                algorithmic_bytes_per_unit=int(unit), n=int(n), m=int(m), nnz=int(nnz_s),
                single_panel_us_per_launch=round(ms1 * 1e3, 2),
                single_panel_frac=round(unit / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))


def gram_mfma(ctx, nv, c, reps=20):
    """K5: G = Z^T Z on v_mfma_f64_16x16x4_f64 -- the 2*NV*c^2 flops of the compression step
     against the FP64 matrix peak (78.6 TFLOP/s)."""
    import torch
    z = torch.randn(nv, c, dtype=torch.float64, device="cuda")
    g = torch.empty(c, c, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ctx.time_gram_dev(z.data_ptr(), c, g.data_ptr(), 3)
    ms = min(ctx.time_gram_dev(z.data_ptr(), c, g.data_ptr(), reps) for _ in range(3))
    flops = 2.0 * nv * c * c
    tf = flops / (ms * 1e-3) / 1e12
    return dict(bound="mfma", achieved=round(tf, 2), peak=78.6, unit="TFLOP/s",
                frac=round(tf / 78.6, 4), kernel="ricadi::gemm_tn_kernel<4,4> (symmetric)",
                us_per_launch=round(ms * 1e3, 1), nv=int(nv), c=int(c))


def cpu_baseline(pr, ms, m, adi_steps, nsample_shifts=3, solves_per_shift=4):
    """Oracle (scipy SuperLU, as the reference's technology) on a bounded sample:
    LU of `nsample_shifts` shifted saddle matrices + `solves_per_shift` panel solves
    each; extrapolated to the step's work (16 LUs + adi_steps shift-solves)."""
    from oracle import lin_alg_utils as olau
    calA = (-pr.A - pr.Nc).T.tocsr()
    rng = np.random.default_rng(0)
    R = rng.standard_normal((pr.NV, m))
    t_lu, t_solve, nlu, nsol = 0.0, 0.0, 0, 0
    pick = [ms[0], ms[len(ms) // 2], ms[-1]][:nsample_shifts]
    for p in pick:
        t0 = time.perf_counter()
        lu = olau.SaddleLU(calA + p * pr.M, pr.J)
        t_lu += time.perf_counter() - t0
        nlu += 1
        for _ in range(solves_per_shift):
            t0 = time.perf_counter()
            lu.solve(R)
            t_solve += time.perf_counter() - t0
            nsol += 1
    lu_s, sol_s = t_lu / nlu, t_solve / nsol
    step_time = len(ms) * lu_s + adi_steps * sol_s
    return dict(value=round(adi_steps / step_time, 3), unit="shift-solves/s", cores=1,
                kind="port",
                sample="%d sparse LUs (%.2f s each) + %d panel solves of m=%d (%.3f s each) "
                       "of the cfg2 saddle matrix, scipy SuperLU single-threaded; "
                       "extrapolated to one step = %d LUs + %d shift-solves"
                       % (nlu, lu_s, nsol, m, sol_s, len(ms), adi_steps),
                lu_seconds=round(lu_s, 3), solve_seconds=round(sol_s, 4))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--N", type=int, default=58, help="mesh parameter (58 = BASELINE cfg2)")
    ap.add_argument("--nu", type=float, default=0.05)
    ap.add_argument("--shifts", type=int, default=16)
    ap.add_argument("--sequential", action="store_true",
                    help="N=1 only: the Newton step through ricadi_ric_newtonadi (the drop-in's C++ "
                         "path), ADI steps one at a time")
    ap.add_argument("--cpp-sweeps", action="store_true",
                    help="with --sequential: the same C++ path in sweep form (sweep_width = "
                         "--sweep-width), as optconpy_amd.proj_ric_utils.proj_alg_ric_newtonadi runs it")
    ap.add_argument("--sweep-width", type=int, default=16,
                    help="shifts per sweep (<= 16; default: the whole 16-shift cycle in one sweep)")
    ap.add_argument("--streams", type=int, default=1,
                    help="1 (default): the shifts of a rank go through one batched solve; k > 1: they "
                         "are cut into k batches that run concurrently (one library context, HIP "
                         "stream and host thread each; measured at cfg2: 2 -> +5 %, 3 -> +7 %, 4 -> "
                         "+6 %, 8 -> -12 % against one batch of 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large-roofline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from optconpy_amd import _lib, backend, problems as pb

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0:
        log("note: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world))
    os.environ["RICADI_DEVICE"] = str(local)

    t0 = time.time()
    pr, tb, trct, ms = build_inputs(args.N, args.nu, args.shifts)
    backend.reset()
    calA = (-pr.A - pr.Nc).T.tocsr()
    calE = pr.M.T.tocsr()
    # developer hook for option sweeps: RICADI_OPTS="agg_v=24,agg_p=36,gmres_restart=20"
    xopts = {}
    for kv in filter(None, os.environ.get("RICADI_OPTS", "").split(",")):
        k, v = kv.split("=")
        xopts[k] = float(v) if "tol" in k else int(v)
    ctx = _lib.Context(local, **xopts)
    ctx.set_operator(calA, calE, pr.J)
    d = dict(pb.default_nwtn_adi_dict(), ms=ms)
    prm_full = _lib.adi_params(dict(d, sweep_width=16))     # untimed reference solve: sweep form
    prm_one = _lib.adi_params(dict(d, nwtn_max_steps=1,
                                   compress_cols=int(os.environ.get("RICADI_CC", "0")),
                                   sweep_width=args.sweep_width if args.cpp_sweeps else 1))
    nb, mw = tb.shape[1], trct.shape[1]
    m = nb + mw
    n = pr.NV + pr.NP
    nnz_s = (calA + calE).nnz + 2 * pr.J.nnz

    # untimed: converge the Newton iteration once; its compressed iterate is the
    # linearisation point of the timed Newton step (so that the step sees the
    # closed-loop low-rank term and the full m = 16 panel, like steps >= 2 do)
    Zfull, info_full = ctx.ric_newtonadi(ms, tb, trct, prm_full, fetch=False)
    Zk = ctx.factor_get()          # the Newton driver leaves the compressed iterate on the device
    K_ref = -ctx.gain(tb)
    if rank == 0:
        log("setup %.1fs: n=%d nnz(S)=%d m=%d; converged Newton: %s; |Z_k| cols %d"
            % (time.time() - t0, n, nnz_s, m, info_full, Zk.shape[1]))

    use_sp = world > 1 or not args.sequential
    if not use_sp:
        def one_step():
            ctx.clear_cache()                       # per-shift setup is part of the step
            _, info = ctx.ric_newtonadi(ms, tb, trct, prm_one, Z0=Zk, fetch=False)
            K = -ctx.gain(tb)
            return info["shift_solves"], info["gmres_iters"], K, info["shift_solves"]
    else:
        from optconpy_amd.shift_parallel import HipOps, lyap_adi_shift_parallel
        nstreams = max(1, args.streams)
        extra = []
        for _ in range(max(0, nstreams - 1)):
            cx = _lib.Context(local, **xopts)
            cx.set_operator(calA, calE, pr.J)
            extra.append(cx)
        ops = HipOps(ctx, extra)
        G = max(1, min(args.sweep_width, 16, len(ms)))
        # closed-loop operator cal A - K_k B^T and rhs [W, K_k] of the Newton step
        Kk = -K_ref                                  # K_k = E Z_k Z_k^T B
        from optconpy_amd import lin_alg_utils as lau
        Wp = lau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=trct, transposedprj=True)
        backend.reset()
        rhs = ops.to_panel(np.hstack([Wp, Kk]))
        tbd = ops.to_panel(tb)

        def one_step():
            ops.clear_cache()
            if not os.environ.get('BENCH_NO_LR'):     # developer probe: open-loop operator
                ops.set_lowrank(Kk, tb)
            ops.gmres_iters = 0
            ops.shift_solves = 0
            blocks, info = lyap_adi_shift_parallel(ops, ms, rhs, adi_max_steps=200,
                                                   adi_newZ_reltol=1e-8, width=G)
            ops.set_lowrank(None, None)
            Z = torch.cat(blocks, dim=1).contiguous()
            Kt = ops.gain(-1.0, Z, tbd)             # gain on the replicated factor
            return info["adi_steps"], ops.gmres_iters, Kt.cpu().numpy(), ops.shift_solves

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    if use_sp:
        ops.t_solve = 0.0

    def ops_t_solve():
        return ops.t_solve
    t0 = time.perf_counter()
    units = 0
    iters = 0
    local_solves = 0
    K = None
    for _ in range(args.steps):
        u, it, K, ls = one_step()
        units += u
        iters += it
        local_solves += ls
    barrier()
    elapsed = time.perf_counter() - t0
    if use_sp and rank == 0:
        log("[bench] rank 0: %.1f %% of the timed region inside the batched shift-solves "
            "(incl. per-shift setup)" % (100.0 * ops_t_solve() / max(elapsed, 1e-9)))
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    k_err = float(np.linalg.norm(K - K_ref) / np.linalg.norm(K_ref))

    if rank == 0:
        out = {
            "metric": "ADI shift-solves/sec (wall-clock to feedback K in ms_per_step)",
            "value": round(units / elapsed, 3),
            "unit": "shift-solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2),
            "higher_is_better": True,
            "scaling": "strong",          # the same problem at every N
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "driven cavity N=%d (cfg2: n=%d, nnz(S)=%d), nu=%g, %d log-spaced "
                            "ADI shifts, 1 Newton step (closed-loop Lyapunov ADI to "
                            "adi_newZ_reltol=1e-8) + gain K; panel m=%d; GMRES tol 1e-10"
                            % (args.N, n, nnz_s, args.nu, len(ms), m),
                "shift_solves_per_step": units // args.steps,
                "gmres_iters_per_shift_solve": round(iters / max(local_solves, 1), 1),
                "parallelism": ("C++ Newton-ADI, %s, 1 GPU" % ("sweeps of %d shifts in one batched solve"
                                                              % args.sweep_width if args.cpp_sweeps
                                                              else "ADI steps one at a time")) if not use_sp
                else "shift-parallel ADI on %d GPU(s), %d shifts/sweep, %s, 1 all-gather/sweep"
                % (world, G, "one batched lockstep solve per rank and sweep" if nstreams == 1
                   else "%d concurrent batched solves per rank and sweep" % nstreams),
                "K_rel_diff_vs_converged": k_err,
                "storage": "arithmetic and all residual checks FP64; Krylov basis stored in %s, "
                           "preconditioner inverses in %s (RICADI_BASIS64=1 RICADI_PRECOND64=1: FP64 storage)"
                           % ("FP64" if os.environ.get("RICADI_BASIS64") else
                              "FP32" if os.environ.get("RICADI_BASIS32") else "FP16",
                              "FP64" if os.environ.get("RICADI_PRECOND64") else "FP32"),
            },
        }
        # the launch of the hot path: the sweep's shifts of one rank in one batched launch
        gsh = [float(p) for p in ms[:max(1, (G if use_sp else 1) // world)]]
        out["roofline"] = spmm_roofline(ctx, nnz_s, n, m, gsh)
        out["roofline_gram_mfma"] = gram_mfma(ctx, pr.NV, 512)
        if world == 1 and not args.no_large_roofline:
            # the HBM-resident instance (BASELINE cfg5 pattern, n ~ 5e5)
            try:
                prl = pb.ricc_problem(236, 0.05, with_convection=True)
                cl = _lib.Context(local, coarse_max=2048)
                cl.set_operator((-prl.A - prl.Nc).T.tocsr(), prl.M.T.tocsr(), prl.J)
                nl = prl.NV + prl.NP
                nnzl = (prl.A + prl.Nc + prl.M).nnz + 2 * prl.J.nnz
                out["roofline_cfg5"] = spmm_roofline(cl, nnzl, nl, 16, gsh, reps=50)
                cl.close()
            except Exception as e:                   # never lose the headline line
                out["roofline_cfg5"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pr, ms, m, units // args.steps)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
