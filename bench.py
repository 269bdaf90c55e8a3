#!/usr/bin/env python
"""Benchmark of the low-rank Newton-ADI hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): ADI shift-solves per second, next to the wall-clock to
the feedback gain K.  One *step* = one Newton step of the projected Riccati
solve on BASELINE config 2 (driven cavity, N=58 -> n = 29 930, nu = 0.05,
16 log-spaced ADI shifts, right-hand-side panel m = NY' + NU = 16), started from
the converged iterate (so that the step sees the closed-loop low-rank term and the
full m = 16 panel, like every Newton step but the first):

    Z = pru.proj_alg_ric_newtonadi(mmat=M, amat=-A-N, jmat=J, bmat=B~, wmat=C~^T, z0=Z_k,
                                   nwtn_adi_dict={..., nwtn_max_steps: 1})['zfac']
    K = -pru.get_mTzzTtb(M^T, Z, B~)

-- the reference's own calls (optcont_main.py:488-492,505) resolved to this repo's
drop-in package `sadptprj_riclyap_adi`, i.e. THROUGH THE BOUNDARY: closed-loop
low-rank ADI to adi_newZ_reltol = 1e-8 (optcont_main.py:124) in sweeps of 16 shifts
(one batched lockstep GMRES per sweep), Newton update norm, recompression, gain.
One *unit* = one shift-solve: one saddle-point solve S(p) [V;L] = [R;0] with an
NV x 16 panel to relative residual 1e-10.  Per-shift setup (the counterpart of the
reference's sparse LUs) is part of every step: the cache is cleared first.

N > 1 (one process per GPU, torch.distributed over RCCL): THE SAME STEP through the same
boundary calls -- under torch.distributed the drop-in shards the ADI sweeps by shift over
the ranks inside the library (ricadi_set_exchange: fixed owner per shift, every rank sets up
and solves only its own shifts, one all-gather of the solution panels per sweep;
projection, recombination, recompression, update norm and gain replicated) -> "scaling":
"strong".  `python bench.py --gpus N` without WORLD_SIZE starts its own N ranks
(`python -m torch.distributed.run ... bench.py` as a CHILD process, before anything touches
the GPU) and relays rank 0's JSON line.

The JSON line also carries: the same step through the Python sweep driver
(`value_python_sweep_driver`), with FP64-stored Krylov basis / preconditioner
(`value_fp64_storage`), roofline objects for K1 (the saddle SpMM: median AND best of
the trials, per-panel and batched-form byte models) and for the kernels that dominate
the run time (block-Jacobi sweep, coarse apply, the three Arnoldi kernels), the Gram
and TSQR MFMA figures, and the CPU baseline (oracle = scipy SuperLU, the full step
on the host cores).  Every figure's kernel duration is measured live with HIP events
on the library's stream; the committed rocprofv3 summaries are under profiles/.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_MFMA_PEAK_TF = 78.6


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_inputs(N, nu, nshifts, pmax=3e3, interleave=False):
    """cfg2 inputs, prepared with the drop-in's own modules (GPU solves), as
    optcont_main.py:405-425 prepares them."""
    import sadptprj_riclyap_adi.lin_alg_utils as lau
    from optconpy_amd import problems as pb
    pr = pb.ricc_problem(N, nu, NU=4, NY=4, alphau=1e-2)
    mct = lau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=pr.mc_mat.T, transposedprj=True)
    tb = lau.apply_invsqrt_fromright(pr.rmat, pr.b_mat, output="dense")
    trct = lau.apply_invsqrt_fromright(pr.y_masmat, mct, output="dense")
    ms = pb.logshifts(1.0, pmax, nshifts, interleave=interleave)
    return pr, tb, trct, ms


def oracle_gain(N, nu, ms):
    """K of the CPU oracle for this workload from the committed fixtures (cfg2: N = 58; cfg3 / cfg3i: N = 75 with
    the 32 shifts in ascending / interleaved order): the fixture whose mesh, viscosity and shift LIST match."""
    for name in ("cfg2_golden.npz", "cfg3i_golden.npz", "cfg3_golden.npz"):
        path = os.path.join(ROOT, "tests", "golden", name)
        if not os.path.exists(path):
            continue
        g = np.load(path)
        cfg = g["cfg"]
        if int(cfg[0]) == N and abs(cfg[1] - nu) <= 1e-15 and len(g["shifts"]) == len(ms) and \
                np.allclose(g["shifts"], ms, rtol=1e-14):
            return g["K_ric"]
    return None


def trials(fn, ntrial=5):
    """Median and best of `ntrial` measurements (ms)."""
    v = [fn() for _ in range(ntrial)]
    return statistics.median(v), min(v)


def traffic_entry(key):
    for name in ("r04_spmm_traffic.json", "r03_spmm_traffic.json", "r02_spmm_traffic.json", "r01_spmm_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                e = json.load(f).get(key)
            if e:
                return e.get("hbm_bytes"), name
        except (OSError, ValueError):
            pass
    return None, None


def spmm_roofline(ctx, nnz_k, nnz_j, n, m, shifts, reps=200):
    """K1 roofline on the launch the hot path issues: ONE batched tile-SpMM launch over the
    G = len(shifts) panels of a sweep (grid.z = G), HIP events on the ctx stream.
    Two byte models, both reported:
      per-panel   G x B_spmm,  B_spmm  = 12 nnz(S) + 4 (n+1) + 16 n m           (`frac`)
      batched     B_batch = nnz(K) (8+8+4) + 2 nnz(J) 12 + 4 (n+1) + 16 n m G   (`frac_batched_form`)
    -- the second counts the matrix once for all G shifts.  `frac` uses the MEDIAN of five
    trials of `reps` launches; the best trial is next to it."""
    import torch
    G = len(shifts)
    nnz_s = nnz_k + 2 * nnz_j
    x = torch.randn(G, n, m, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    al, be = [float(p) for p in shifts], [1.0] * G
    ctx.time_spmm_batch_dev(al, be, x.data_ptr(), m, y.data_ptr(), 20)          # warm-up
    med, best = trials(lambda: ctx.time_spmm_batch_dev(al, be, x.data_ptr(), m, y.data_ptr(), reps))
    sinfo = ctx.setup_info()
    variant = sinfo.get("k1_variant", -1)          # kernel of the batched launch (the library's own record)
    xb = 4.0 if variant >= 4 else 8.0               # panel bytes per entry as the launch moved them: FP32 Z_j in ...
    yb = 4.0 if sinfo.get("fp32_operator_output", 0) == 1 else 8.0     # ... FP32 w out (round 4), else FP64
    med1, best1 = trials(lambda: ctx.time_spmm_dev(al[0], 1.0, x.data_ptr(), m, y.data_ptr(), reps))
    unit = 12.0 * nnz_s + 4.0 * (n + 1) + 16.0 * n * m
    nbytes = unit * G
    b_batch = nnz_k * 20.0 + 2.0 * nnz_j * 12.0 + 4.0 * (n + 1) + 16.0 * n * m * G
    gbs = nbytes / (med * 1e-3) / 1e9
    traffic, tsrc = traffic_entry("%dx%dx%d" % (G, n, m))
    # which kernel served the launch: the library says (ricadi_setup_info slot 18), nothing is guessed here
    ms = (variant & 3) == 2
    kname = {0: "ricadi::spmm_kernel_v2 (CSR)", 1: "ricadi::spmm_blocked_kernel", 2: "ricadi::spmm_blocked_ms_kernel"}.get(
        variant & 3, "unknown")
    # `frac`: the launch serves G shifts of ONE pattern, so the bytes that HAVE to move are the batched form's
    # (matrix once for all shifts) whatever the kernel streams; a single-shift launch is the per-panel
    # form.  The per-panel reading of the same launch (each shift's assembled value array counted) is kept beside it.
    model = b_batch if G > 1 else nbytes
    gbs_model = model / (med * 1e-3) / 1e9
    return dict(bound="hbm", achieved=round(gbs_model, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(gbs_model / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=tsrc,
                byte_model=("batched form: K and M values + pattern once for all %d shifts of the launch, panels in/out "
                            "per shift" % G) if G > 1 else "per panel (B_spmm)",
                frac_per_panel_model=round(gbs / HBM_PEAK_GBS, 4),
                kernel=kname, kernel_reads=("one value set for all groups (18 B per non-zero)" if ms else
                                            "one assembled value array per group (10 B per non-zero and group)"),
                us_per_launch=round(med * 1e3, 2),
                us_per_launch_best=round(best * 1e3, 2),
                frac_best=round(model / (best * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                units_per_launch=G, algorithmic_bytes=int(model),
                algorithmic_bytes_per_unit=int(unit),
                algorithmic_bytes_per_panel_model=int(nbytes),
                algorithmic_bytes_batched_form=int(b_batch),
                frac_batched_form=round(b_batch / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                panels_as_stored="x %s in, y %s out (the algorithmic bytes count FP64 panels, SURVEY.md 8d)"
                                 % ("FP32" if xb == 4.0 else "FP64", "FP32" if yb == 4.0 else "FP64"),
                bytes_batched_form_as_stored=int(b_batch - (16.0 - xb - yb) * n * m * G),
                frac_batched_form_as_stored=round((b_batch - (16.0 - xb - yb) * n * m * G) / (med * 1e-3) / 1e9
                                                  / HBM_PEAK_GBS, 4),
                traffic_over_batched_form=(round(traffic / b_batch, 3) if traffic else None),
                n=int(n), m=int(m), nnz=int(nnz_s),
                single_panel_us_per_launch=round(med1 * 1e3, 2),
                single_panel_frac=round(unit / (med1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                timing="median of 5 trials x %d launches, HIP events" % reps)


def kernel_rooflines(ctx, shifts, m, nvec=7, reps=100):
    """Roofline objects for the kernels that dominate the run time besides K1, on the launch
    the batched GMRES issues (G groups).  Algorithmic bytes per launch:
      block_apply<32,float>   G (4 nb 32^2 + 16 nv m)         FP32 inverses + panel in/out
      dense_apply_tiled       G (4 (16 ceil(k/16))^2 + 16 k m)  FP32 coarse inverse + rc/ec
      cols_dots               G ((nvec b + 8) n m)            nvec basis vectors of b bytes/entry + w
      cols_update_dots        G ((nvec b + 16) n m)           basis (second pass from cache) + w in/out
      cols_update             G ((nvec b + 16 + b) n m)       basis + w in, FP64 + stored copy out
    with b = 2 (FP16-stored basis; 4 / 8 with RICADI_BASIS32 / 64)."""
    G = len(shifts)
    al, be = [float(p) for p in shifts], [1.0] * G
    n, nv = ctx.n, ctx.nv
    info = ctx.setup_info()
    nb, kc = info["nbv"], info.get("dense_coarse", info["kc"])   # the dense inverse of the LAST preconditioner level
    b = 8 if os.environ.get("RICADI_BASIS64") else (4 if os.environ.get("RICADI_BASIS32") else 2)
    pb_ = 8 if os.environ.get("RICADI_PRECOND64") else 4
    kp = 16 * ((kc + 15) // 16)
    ctx.time_kernel_dev("dots", al, be, m, nvec=nvec, reps=1)       # the library then knows which panel the passes read
    wb = 4.0 if ctx.setup_info().get("fp32_operator_output", 0) == 1 else 8.0
    models = {
        # a plain velocity-sized sweep: not launched by the folded cycle (see pc_two_term / pc_rect below); kept as the
        # reference point of the block-Jacobi apply itself
        "block_v": ("ricadi::block_apply_kernel<32,%s> (plain sweep; not on the folded hot path)"
                    % ("double" if pb_ == 8 else "float"), G * (pb_ * nb * 1024.0 + 16.0 * nv * m)),
        "coarse": ("ricadi::dense_apply_tiled_kernel" if pb_ == 4 else "ricadi::dense_apply_kernel<double>",
                   G * (pb_ * float(kp) * kp + 16.0 * kc * m)),
        # w: 4 B per entry where the operator writes the FP32 panel (round 4), read once per pass; the second pass keeps
        # w as it is on the FP16 16-column path (nothing written back), else it writes the projected w
        "dots": ("ricadi::cols_dots%s_kernel (+reduce_partials)" % ("16" if b == 2 and m == 16 else ""),
                 G * ((nvec * b + wb) * n * m)),
        "update_dots": ("ricadi::cols_update_dots%s_kernel (+reduce_partials)" % ("16" if b == 2 and m == 16 else ""),
                        G * ((nvec * b + wb + (0.0 if b == 2 and m == 16 else 8.0)) * n * m)),
        # ... + w read + the new vector in storage precision (+ its FP64 copy unless the preconditioner reads the FP16 one)
        "update": ("ricadi::cols_update16_hess_kernel (last Arnoldi pass + Hessenberg / Givens update)" if b == 2 and m == 16
                   else "ricadi::cols_update_kernel",
                   G * ((nvec * b + wb + b + (0.0 if b == 2 and info.get("fp16_vector_input") else 8.0)) * n * m)),
    }
    out = {}
    for key, (kname, nbytes) in models.items():
        ctx.time_kernel_dev(key, al, be, m, nvec=nvec, reps=10)
        med, best = trials(lambda: ctx.time_kernel_dev(key, al, be, m, nvec=nvec, reps=reps))
        gbs = nbytes / (med * 1e-3) / 1e9
        out[key] = dict(bound="hbm", kernel=kname, achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(gbs / HBM_PEAK_GBS, 4), us_per_launch=round(med * 1e3, 2),
                        us_per_launch_best=round(best * 1e3, 2), algorithmic_bytes=int(nbytes),
                        units_per_launch=G, nvec=nvec if key in ("dots", "update_dots", "update") else None)
    # The stages of ONE preconditioner application, each issued alone by the solver's own code path
    # (ricadi_time_kernel_dev 10 + k), so kernel and template instance are the ones the iteration launches
    # at this size.  Byte models per launch; the FP32-stored operands are pb_ bytes.
    kc0, npn = info["kc"], info["np"]
    rk, tk = info.get("rect_ks", 0), info.get("two_term_ks", 0)
    h16 = bool(info.get("fp16_vector_input")) and b == 2 and m <= 16
    vin = 2.0 if h16 else 8.0
    fl = "float" if pb_ == 4 else "double"
    few = nb * G <= 8192
    # round 4: the velocity part between the three sweeps is an FP32 panel where the operator reads the FP32 Z_j anyway
    # (reported by the library after the first timed application); the timed stages then neither write nor read an
    # FP64 z
    ctx.time_kernel_dev("precond", al, be, m, nvec=nvec, reps=1)
    mid = ctx.setup_info().get("fp32_intermediate", 0) == 1
    zb = 4.0 if mid else 8.0
    hot = m == 16 and info["bs"] == 32 and os.environ.get("RICADI_SWEEP_META", "1") != "0"   # record-driven sweep kernels
    # ... which read BF16-stored blocks (2 B per entry) in the FP32 cycle
    pbb = 2.0 if (hot and mid and pb_ == 4 and os.environ.get("RICADI_BLOCKS16", "1") != "0") else pb_
    bl = "unsigned short (BF16)" if pbb == 2.0 else fl
    w64 = 0.0 if (mid or ctx.setup_info().get("k1_variant", 0) >= 4) else 8.0    # FP64 z stored by the last sweep?
    stages = {
        "pc_restrict": (("ricadi::spmm_rowwave_kernel" if m == 16 and info.get("nnz_restriction", n) >= 32 * max(kc0, 1)
                         else "ricadi::spmm_kernel_v2")
                        + (" (rows of P^T: smoothed aggregation, %.1f entries per dof)"
                           % (info.get("nnz_restriction", n) / max(n, 1)) if info.get("nnz_restriction", n) > n
                           else " (unit values, aggregate lists)"),
                        G * (vin * n * m + 8.0 * kc0 * m) + (12.0 if info.get("nnz_restriction", n) > n else 4.0)
                        * info.get("nnz_restriction", n)),
        "pc_coarse": (models["coarse"][0] if info["levels"] <= 2 else "child level: one full cycle of its own stages",
                      models["coarse"][1] if info["levels"] <= 2 else None),
        "pc_sy_prows": ("ricadi::spmm_kernel_v2 (pressure rows of S*Y)", None),
        "pc_two_term": (("ricadi::block_two32_kernel<%d,%s,%s>" if hot else "ricadi::block_apply2_kernel<32,%d,%s,%s>")
                        % (tk, bl if hot else fl, "true" if h16 else "false") if tk else "ricadi::block_apply_kernel<32,%s>" % fl,
                        G * (pbb * nb * 32.0 * (32 + tk) + (vin + zb) * nv * m + 8.0 * kc0 * m)),
        "pc_jprod": ("ricadi::spmm_kernel_v2 (J)", 12.0 * info.get("nnz_j", 0) + G * (8.0 * nv * m + 16.0 * npn * m)),
        "pc_schur": (("ricadi::block_apply_rect_kernel<32,32,%s> (block's own rows as input list)" % fl)
                     if info["nbp"] * G <= 8192 else "ricadi::block_apply_kernel<32,%s>" % fl,
                     G * (pb_ * info["nbp"] * 1024.0 + (8.0 + 8.0 + 8.0 + 4.0) * npn * m)),
        "pc_rect": (("ricadi::block_rect32_kernel<%d,%s,%s>" % (rk, bl, "true" if mid else "false") if hot else
                     "ricadi::block_apply_rect_kernel<32,%d,%s>" % (rk, fl)) if rk
                    else "ricadi::block_apply_kernel<32,%s> (+ CSR J^T input)" % fl,
                    G * (pbb * nb * 32.0 * max(rk, 32) + (zb + w64) * nv * m + 4.0 * n * m + 8.0 * npn * m
                         + 8.0 * kc0 * m)),
    }
    if m == 16 and info["bs"] == 32 and npn > 0:
        # K2p: the three launches of the pressure step are ONE kernel; its bytes: J once (shared by the groups), per
        # group the Schur block inverses, the gathered velocity rows of z, r_p, the coarse correction and z_p (+ plain
        # copy for the J^T product and the FP32 copy)
        stages["pc_sy_prows"] = ("(fused into pc_schur: ricadi::pressure_step_kernel)", None)
        stages["pc_jprod"] = ("(fused into pc_schur: ricadi::pressure_step_kernel)", None)
        stages["pc_schur"] = ("ricadi::pressure_step_kernel<%s,%s,%s>" % (bl, "_Float16" if h16 else "double",
                                                                           "float" if mid else "double"),
                              12.0 * info.get("nnz_j", 0) + G * (pbb * info["nbp"] * 1024.0 + zb * nv * m +
                                                                   (vin + 20.0) * npn * m + 8.0 * kc0 * m))
    tot = 0.0
    for key, (kname, nbytes) in stages.items():
        try:
            ctx.time_kernel_dev(key, al, be, m, nvec=nvec, reps=10)
            med, best = trials(lambda: ctx.time_kernel_dev(key, al, be, m, nvec=nvec, reps=reps), 3)
        except Exception as e:
            out[key] = {"error": str(e)}
            continue
        tot += med
        o = dict(kernel=kname, us_per_launch=round(med * 1e3, 2), us_per_launch_best=round(best * 1e3, 2),
                 units_per_launch=G)
        if nbytes:
            gbs = nbytes / (med * 1e-3) / 1e9
            o.update(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                     frac=round(gbs / HBM_PEAK_GBS, 4), algorithmic_bytes=int(nbytes))
        out[key] = o
    out["precond_stages_sum_us"] = round(tot * 1e3, 1)
    # anatomy of one lockstep iteration at G groups (sum over its launches)
    ctx.time_kernel_dev("precond", al, be, m, nvec=nvec, reps=5)
    med, _ = trials(lambda: ctx.time_kernel_dev("precond", al, be, m, nvec=nvec, reps=50), 3)
    out["precond_apply_all_launches_us"] = round(med * 1e3, 1)
    return out


def gram_mfma(ctx, nv, c, reps=20):
    """K5 (default compression route): G = Z^T Z on v_mfma_f64_16x16x4_f64 -- 2 NV c^2 flops
    against the FP64 matrix peak."""
    import torch
    z = torch.randn(nv, c, dtype=torch.float64, device="cuda")
    g = torch.empty(c, c, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ctx.time_gram_dev(z.data_ptr(), c, g.data_ptr(), 3)
    med, best = trials(lambda: ctx.time_gram_dev(z.data_ptr(), c, g.data_ptr(), reps))
    flops = 2.0 * nv * c * c
    tf = flops / (med * 1e-3) / 1e12
    return dict(bound="mfma", achieved=round(tf, 2), peak=FP64_MFMA_PEAK_TF, unit="TFLOP/s",
                frac=round(tf / FP64_MFMA_PEAK_TF, 4), kernel="ricadi::gemm_tn_kernel<4,4> (symmetric)",
                us_per_launch=round(med * 1e3, 1), us_per_launch_best=round(best * 1e3, 1),
                nv=int(nv), c=int(c))


def tsqr_mfma(ctx, nv, c, reps=3):
    """K5 (tall-skinny QR, north_star's "MFMA utilisation on the TSQR"): thin QR of an NV x c factor
    by ricadi_qr's device path -- 128-column panels by CholQR2 on the MFMA GEMMs (the 128 x 128 Cholesky
    factor and its inverse in one workgroup; Householder TSQR tree on 32-column panels when a panel is
    too ill-conditioned for it) inside a block Gram-Schmidt with
    re-orthogonalisation, also on the MFMA GEMMs.  Algorithmic flops 2 NV c^2 - 2/3 c^3
     over the whole factorisation's duration (HIP events)."""
    import torch
    z = torch.randn(nv, c, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ctx.time_qr_dev(z.data_ptr(), c, 1)
    med, best = trials(lambda: ctx.time_qr_dev(z.data_ptr(), c, reps), 3)
    flops = 2.0 * nv * c * c - (2.0 / 3.0) * c ** 3
    tf = flops / (med * 1e-3) / 1e12
    return dict(bound="mfma", achieved=round(tf, 3), peak=FP64_MFMA_PEAK_TF, unit="TFLOP/s",
                frac=round(tf / FP64_MFMA_PEAK_TF, 4),
                kernel="ricadi block QR: 128-column CholQR2 panels (gemm_tn / gemm_nn on MFMA + cholqr_wide_kernel) "
                       "inside block Gram-Schmidt (gemm_tn / gemm_nn); Householder TSQR tree as fallback",
                ms_per_factorisation=round(med, 2), ms_best=round(best, 2), nv=int(nv), c=int(c))


def _cpu_shift_worker(args):
    """One process per shift (BASELINE.md plan (b)): LU of its shifted saddle matrix once, then
    `nsolve` panel solves -- the work one shift contributes to a step in sweep form."""
    calA, M, J, p, m, nsolve, seed = args
    from oracle import lin_alg_utils as olau
    t0 = time.perf_counter()
    lu = olau.SaddleLU(calA + p * M, J)
    t_lu = time.perf_counter() - t0
    R = np.random.default_rng(seed).standard_normal((M.shape[0], m))
    t0 = time.perf_counter()
    for _ in range(nsolve):
        lu.solve(R)
    return t_lu, time.perf_counter() - t0


def cpu_baseline_sample(calA, M, J, ms, m, units, npick=2, label="", n_lu=None):
    """Bounded single-core sample of the oracle (scipy SuperLU, the reference's technology) for the larger
    workloads: `npick` of the sparse LUs and one panel solve on each, LU and solve time reported separately
    (the reference amortises one LU per shift over the ADI cycles), priced up to the line's step of
    `units` shift-solves over len(ms) shifts."""
    from oracle import lin_alg_utils as olau
    R = np.random.default_rng(0).standard_normal((M.shape[0], m))
    ns = len(ms)
    pick = [ms[(k * (ns - 1)) // max(npick - 1, 1)] for k in range(npick)] if npick > 1 else [ms[ns // 2]]
    t_lu = t_sol = 0.0
    for p in pick:
        t0 = time.perf_counter()
        lu = olau.SaddleLU(calA + p * M, J)
        t_lu += time.perf_counter() - t0
        t0 = time.perf_counter()
        lu.solve(R)
        t_sol += time.perf_counter() - t0
        del lu
    lu_s, sol_s = t_lu / len(pick), t_sol / len(pick)
    n_lu = ns if n_lu is None else n_lu
    step_time = n_lu * lu_s + units * sol_s
    return dict(value=round(units / step_time, 3), unit="shift-solves/s", cores=1, kind="port",
                sample="%d sparse LUs (%.2f s each) + %d panel solves of m=%d (%.3f s each) of the %s saddle matrices "
                       "(n = %d), scipy SuperLU single-threaded; priced up to one step = %d LUs + %d shift-solves "
                       "(%.0f s)" % (len(pick), lu_s, len(pick), m, sol_s, label, M.shape[0] + J.shape[0], n_lu, units,
                                     step_time),
                lu_seconds=round(lu_s, 3), solve_seconds=round(sol_s, 4), step_seconds=round(step_time, 2))


def cpu_baseline(pr, ms, m, adi_steps, full=False):
    """Oracle (scipy SuperLU -- the reference's technology) on the host cores, same matrices.
    One step = len(ms) sparse LUs + adi_steps shift-solves (the reference amortises one LU per
    shift over the ADI cycles).
      `value` (cores = 1): bounded sample -- 4 of the LUs and 2 panel solves on each (~12 s),
         priced up to the step; with --cpu-full ALL LUs and ALL solves are run (~100 s);
      `parallel`: the FULL step, nothing extrapolated, as the sweep form parallelises it on a
         CPU: one process per shift (LU + its adi_steps/len(ms) panel solves), min(cores, 16)
         processes, wall time (BASELINE.md plan (b))."""
    from oracle import lin_alg_utils as olau
    import multiprocessing as mp
    calA = (-pr.A - pr.Nc).T.tocsr()
    M = pr.M.T.tocsr()
    R = np.random.default_rng(0).standard_normal((pr.NV, m))
    ns = len(ms)
    per_shift = max(1, -(-adi_steps // ns))
    pick = list(ms) if full else [ms[0], ms[ns // 3], ms[(2 * ns) // 3], ms[-1]]
    nsol_each = per_shift if full else 2
    t_lu, t_sol = 0.0, 0.0
    for p in pick:
        t0 = time.perf_counter()
        lu = olau.SaddleLU(calA + p * M, pr.J)
        t_lu += time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(nsol_each):
            lu.solve(R)
        t_sol += time.perf_counter() - t0
        del lu
    lu_s, sol_s = t_lu / len(pick), t_sol / (len(pick) * nsol_each)
    step_time = ns * lu_s + ns * per_shift * sol_s
    out = dict(value=round(ns * per_shift / step_time, 3), unit="shift-solves/s", cores=1, kind="port",
               sample=("the full step: %d sparse LUs (%.2f s each) + %d panel solves of m=%d (%.3f s each)"
                       % (ns, lu_s, ns * per_shift, m, sol_s)) if full else
                      ("%d sparse LUs (%.2f s each) + %d panel solves of m=%d (%.3f s each) of this workload's "
                       "saddle matrices (n = %d), scipy SuperLU single-threaded; priced up to one step = %d LUs + "
                       "%d shift-solves (%.0f s)" % (len(pick), lu_s, len(pick) * nsol_each, m, sol_s,
                                                    pr.NV + pr.J.shape[0], ns, ns * per_shift, step_time)),
               lu_seconds=round(lu_s, 3), solve_seconds=round(sol_s, 4), step_seconds=round(step_time, 2))
    try:
        ncore = len(os.sched_getaffinity(0))
    except AttributeError:
        ncore = os.cpu_count() or 1
    nproc = max(1, min(ns, ncore, 16))
    try:
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(nproc) as pool:
            res = pool.map(_cpu_shift_worker, [(calA, M, pr.J, p, m, per_shift, i) for i, p in enumerate(ms)],
                           chunksize=1)
        wall = time.perf_counter() - t0
        out["parallel"] = dict(value=round(ns * per_shift / wall, 3), unit="shift-solves/s", cores=nproc,
                               kind="port",
                               sample="the FULL step, one process per shift on %d cores: %d LUs + %d panel "
                                      "solves, wall %.1f s (sum over processes: LU %.1f s, solves %.1f s)"
                                      % (nproc, ns, ns * per_shift, wall, sum(r[0] for r in res),
                                         sum(r[1] for r in res)),
                               step_seconds=round(wall, 2))
    except Exception as e:                                   # never lose the headline line
        out["parallel"] = {"error": str(e)}
    if "value" in out.get("parallel", {}):
        # the MEASURED figure is the headline of the object: the full step, nothing extrapolated, on the cores
        # actually used; the bounded single-core sample (priced up) stays beside it
        single = {k: out[k] for k in ("value", "unit", "cores", "kind", "sample", "lu_seconds", "solve_seconds",
                                      "step_seconds")}
        par = out["parallel"]
        out = dict(value=par["value"], unit=par["unit"], cores=par["cores"], kind="port", sample=par["sample"],
                   step_seconds=par["step_seconds"], single_core=single)
    return out


# BASELINE.json configs 3-5: fixed-work units for runs at their sizes.  One step =
# ONE pass over the whole shift list of the open-loop projected Lyapunov ADI (s shift-solves with
# an NV x m panel each, sweeps of 16 shifts, Cauchy recombination and residual hand-off included),
# through the boundary call pru.solve_proj_lyap_stein (tests/test_units_compfacres_compress.py:62-64).
# `--workload cfg3` itself is the steady-state Riccati run of cyl_wake_cont.py:34-50 (the Newton step to K
# of the default workload at N = 75, 32 shifts); `cfg3-cycle` is the open-loop pass at that size.
CFG3 = dict(N=75, nu=0.15 / 40.0, shifts=32)
WORKLOADS = {
    "cfg3-cycle": dict(N=75, nu=0.15 / 40.0, shifts=32, m=16, dre=False, pmin=1.0, pmax=3e3,
                 note="cylinder-wake surrogate, n = 50 177, 32 shifts (BASELINE: 4 GPUs)"),
    "cfg4": dict(N=106, nu=0.15 / 60.0, shifts=64, m=66, dre=True, pmin=0.5, pmax=2e3,
                 note="n = 100 490, time-varying DRE operator -(M^T/2 + tau (A+N)^T) at the largest step "
                      "of the sine-squeezed mesh (Nts = 16), panel m = 66, 64 shifts (BASELINE: 8 GPUs)"),
    "cfg5": dict(N=236, nu=0.05, shifts=128, m=16, dre=False, pmin=1.0, pmax=3e3,
                 note="n = 499 850, nnz(S) = 14.4e6, 128 shifts (BASELINE: 8 GPUs)"),
}


def _xopts():
    """Developer hook for option sweeps: RICADI_OPTS="gmres_restart=40,agg_v=24"."""
    xopts = {}
    for kv in filter(None, os.environ.get("RICADI_OPTS", "").split(",")):
        k, v = kv.split("=")
        xopts[k] = float(v) if "tol" in k else int(v)
    return xopts


def cycle_workload(name, world, rank, local, steps, warmup, col_split, barrier_fn, python_driver=False,
                   with_cpu_baseline=True, interleave=True):
    """Times `steps` passes over the shift list of a WORKLOADS entry.  Default: through the boundary call
    pru.solve_proj_lyap_stein (C++ sweep driver; under torch.distributed the library shards the sweeps by shift
    over the ranks, RCCL all-gather on its stream) -- the same call at every rank count.  `python_driver`:
    the Python sweep driver (shift_parallel.py; supports column parts).  Returns the result dict (rank 0)."""
    import torch
    import torch.distributed as dist
    from optconpy_amd import _lib, backend, problems as pb
    from optconpy_amd.shift_parallel import HipOps, lyap_adi_shift_parallel, plan_items
    import sadptprj_riclyap_adi.proj_ric_utils as pru
    w = WORKLOADS[name]
    pr = pb.ricc_problem(w["N"], w["nu"])
    MT = pr.M.T.tocsr()
    if w["dre"]:
        tau = float(np.diff(pb.get_tint(0.0, 1.0, 16, True)).max())
        calA = (-(0.5 * MT + tau * (pr.A.T + pr.Nc.T))).tocsr()
    else:
        calA = (-pr.A - pr.Nc).T.tocsr()
    ms = pb.logshifts(w["pmin"], w["pmax"], w["shifts"], interleave=interleave)
    xo = _xopts()
    if xo:
        backend.configure(**xo)
    ctx = backend.context_for(calA, MT, pr.J) if not python_driver else _lib.Context(local, **xo)
    if python_driver:
        ctx.set_operator(calA, MT, pr.J)
    # right-hand-side factor: seeded random panel, Leray-projected on the device
    R = np.random.default_rng(1234).standard_normal((pr.NV, w["m"]))
    X, _, _ = ctx.shift_solve(1.0, 0.0, R)
    Wh = np.ascontiguousarray(MT @ X[:pr.NV])
    n = pr.NV + pr.NP
    G = min(16, len(ms))
    parts = plan_items(G, world, col_split)
    if python_driver:
        ops = HipOps(ctx)
        W = ops.to_panel(Wh)

        def step():
            ops.clear_cache()
            ops.gmres_iters = ops.shift_solves = ops.nonconverged = 0
            ops.worst_relres = 0.0
            blocks, info = lyap_adi_shift_parallel(ops, ms, W, adi_max_steps=len(ms), adi_newZ_reltol=0.0,
                                                   width=G, col_parts=parts)
            info["gmres_iters"] = ops.gmres_iters
            info["shift_solves"] = ops.shift_solves
            return info
    else:
        dd = dict(pb.default_nwtn_adi_dict(), ms=ms, adi_max_steps=len(ms), adi_newZ_reltol=0.0, sweep_width=G,
                  project_w=False, device_resident=True)

        def step():
            ctx.clear_cache()
            o = pru.solve_proj_lyap_stein(amat=calA, mmat=MT, jmat=pr.J, wmat=Wh, transposed=True, adi_dict=dd)
            return dict(adi_steps=o["adi_steps"], gmres_iters=o["gmres_iters"], shift_solves=o["shift_solves"],
                        gmres_nonconverged=o["gmres_nonconverged"], gmres_worst_relres=o["gmres_worst_relres"],
                        res_fro=o["res_fro"])

    for _ in range(warmup):
        step()
    barrier_fn(ctx)
    t0 = time.perf_counter()
    units = 0
    info = None
    for _ in range(steps):
        info = step()
        units += info["adi_steps"]
    barrier_fn(ctx)
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    sinfo = ctx.setup_info()
    res = dict(workload="%s: driven-cavity pattern N=%d (%s), nu=%g; one step = one pass over the %d-shift list of "
                        "the open-loop Lyapunov ADI, panel m=%d, GMRES tol 1e-10, per-shift setup inside the step; %s"
                        % (name, w["N"], w["note"], w["nu"], len(ms), w["m"],
                           "Python sweep driver" if python_driver else
                           "through pru.solve_proj_lyap_stein (C++ sweep driver%s)"
                           % ("" if world == 1 else ", sweeps sharded by shift inside the library over %d ranks, RCCL "
                              "all-gather on the library's stream" % world)),
               value=round(units / el, 3), unit="shift-solves/s", ms_per_step=round(1e3 * el / steps, 1),
               steps=steps, warmup=warmup, n=int(n), m=int(w["m"]), shifts=len(ms), col_parts=parts,
               shift_order=("interleaved (pb.logshifts(..., interleave=True): any 16 consecutive shifts span the range; "
                            "sweeps of 16)" if interleave and len(ms) > 16 else "ascending |p|"),
               gmres_nonconverged=info["gmres_nonconverged"], gmres_worst_relres=info["gmres_worst_relres"],
               gmres_iters_per_shift_solve=round(info["gmres_iters"] / max(info["shift_solves"], 1), 1),
               final_residual_fro=info["res_fro"],
               preconditioner_levels=sinfo["levels"], dense_coarse_dim=sinfo["dense_coarse"])
    if python_driver:
        ctx.close()
    else:
        backend.reset()
    if with_cpu_baseline and rank == 0 and world == 1:
        try:
            res["cpu_baseline"] = cpu_baseline_sample(calA, MT, pr.J, ms, w["m"], len(ms),
                                                      npick=2 if n < 150000 else 1, label=name)
        except Exception as e:                       # never lose the line
            res["cpu_baseline"] = {"error": str(e)}
    return res


def dre_workload(args, rank):
    """`--workload cfg4-dre`: BASELINE cfg4 as what it is -- the time-varying differential Riccati loop at
    n ~ 1e5 (solve_dae_ric.py:121-211 through optconpy_amd.dae_ric.solve_flow_daeric, the boundary calls
    of every backward time step: Newton-ADI with z0 / w_mat = [M^T Z_c, sqrt(tau) C~^T] (m <= comprz_maxc +
    NY' + NU = 66), compression to comprz_maxc = 50, gain, feed-forward saddle solve), on the sine-squeezed
    time mesh get_tint(0, 1, Nts) (optcont_main.py:141-150) with a NEW operator per step
    (-(M^T/2 + tau_k (A + N(t_k))^T), N(t) = (1 + 0.5 sin(2 pi t)) N(vortex)).  One step = one whole
    backward sweep; per time step wall-clock, Newton / ADI steps and shift-solves are reported."""
    import sadptprj_riclyap_adi.lin_alg_utils as lau
    import sadptprj_riclyap_adi.proj_ric_utils as pru
    from optconpy_amd import backend, problems as pb
    from optconpy_amd.dae_ric import MemoryStore, solve_flow_daeric
    xopts = _xopts()
    if xopts:
        backend.configure(**xopts)
    N = args.N if args.N != 58 else 106
    nu, Nts, ns = 0.15 / 60.0, args.nts, (args.shifts if args.shifts != 16 else 64)
    pr = pb.ricc_problem(N, nu, NU=4, NY=4, alphau=1e-2)
    mct = lau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=pr.mc_mat.T, transposedprj=True)
    tmesh = pb.get_tint(0.0, 1.0, Nts, True)
    nad = dict(pb.default_nwtn_adi_dict(), ms=pb.logshifts(0.5, 2e3, ns, interleave=not args.sorted_shifts))
    NY2 = mct.shape[1]

    def ystar(t):
        return (0.1 * np.sin(5 * np.pi * t) * np.arange(1, NY2 + 1)).reshape(-1, 1)

    def tdpart(time=None, **kw):
        return (1.0 + 0.5 * np.sin(2 * np.pi * time)) * pr.Nc, np.zeros((pr.NV, 1))

    rec = []

    class Timed:
        def __getattr__(self, name):
            f = getattr(pru, name)
            if name != "proj_alg_ric_newtonadi":
                return f

            def g(*a, **k):
                t0 = time.perf_counter()
                o = f(*a, **k)
                rec.append(dict(seconds=round(time.perf_counter() - t0, 3), newton_steps=o["nwtn_steps"],
                                adi_steps=o["adi_steps"], shift_solves=o["shift_solves"],
                                gmres_iters=o["gmres_iters"], rhs_cols=int(k["wmat"].shape[1] + k["bmat"].shape[1]),
                                nonconverged=o["gmres_nonconverged"], escalations=o.get("storage_escalations", 0)))
                log("cfg4-dre: time step %d of %d: %s" % (len(rec), Nts, rec[-1]))
                return o
            return g

    sweep = [0]

    def one():
        sweep[0] += 1
        kw = dict(mmat=pr.M, amat=pr.A, jmat=pr.J, bmat=pr.b_mat, mcmat=mct.T, v_is_my=True, rmat=pr.rmat,
                  vmat=pr.y_masmat, rhsv=np.zeros((pr.NV, 1)), gamma=1e-1, tmesh=tmesh, ystarvec=ystar,
                  nwtn_adi_dict=nad, comprz_thresh=5e-5, comprz_maxc=50, get_tdpart=tdpart,
                  get_datastr=lambda time=None, **k: "dre%d_t%.6f" % (sweep[0], time), gtdtstrargs={})
        del rec[:]
        store = MemoryStore()
        t0 = time.perf_counter()
        fb = solve_flow_daeric(store=store, pru=Timed(), lau=lau, **kw)
        el = time.perf_counter() - t0
        K0 = store.load(fb[tmesh[0]]["mtxtb"])
        return el, list(rec), float(np.linalg.norm(K0))

    for _ in range(args.warmup):
        one()
    tot, steps_rec, k0 = 0.0, None, None
    for _ in range(args.steps):
        el, steps_rec, k0 = one()
        tot += el
    units = sum(r["shift_solves"] for r in steps_rec)
    info = backend.context().setup_info()
    cpu = None
    if not args.no_cpu_baseline:
        try:
            MT = pr.M.T.tocsr()
            tau = float(np.diff(tmesh).max())
            ft = (-(0.5 * MT + tau * (pr.A.T + 1.5 * pr.Nc.T))).tocsr()
            # the operator changes with t: the reference factorises every shifted matrix anew in every time step
            cpu = cpu_baseline_sample(ft, MT, pr.J, list(nad["ms"]), 66, units, npick=2,
                                      label="cfg4-dre (largest time step)", n_lu=ns * Nts)
        except Exception as e:
            cpu = {"error": str(e)}
    return {
        **({"cpu_baseline": cpu} if cpu is not None else {}),
        "metric": "ADI shift-solves/sec (wall-clock of the backward DRE sweep in ms_per_step)",
        "value": round(units * args.steps / tot, 3), "unit": "shift-solves/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * tot / args.steps, 1),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": "cfg4-dre: driven-cavity pattern N=%d (n=%d), nu=%g, differential Riccati sweep over "
                        "get_tint(0,1,%d) (sine-squeezed), %d ADI shifts (interleaved order unless --sorted-shifts), comprz_maxc=50, time-varying "
                        "convection; through solve_flow_daeric -> pru.proj_alg_ric_newtonadi / compress_Zsvd / "
                        "get_mTzzTtb / lau.solve_sadpnt_smw" % (N, pr.NV + pr.NP, nu, Nts, ns),
            "time_steps": steps_rec, "seconds_in_newton_adi": round(sum(r["seconds"] for r in steps_rec), 2),
            "shift_solves_per_sweep": units, "gain_norm_at_t0": k0,
            "preconditioner_levels": info["levels"], "dense_coarse_dim": info["dense_coarse"]}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--N", type=int, default=58, help="mesh parameter (58 = BASELINE cfg2)")
    ap.add_argument("--nu", type=float, default=0.05)
    ap.add_argument("--shifts", type=int, default=16)
    ap.add_argument("--sweep-width", type=int, default=16,
                    help="shifts per sweep (<= 16; default: the whole 16-shift cycle in one sweep)")
    ap.add_argument("--python-driver", action="store_true",
                    help="N=1: time the Python sweep driver (shift_parallel.py, the multi-GPU code path "
                         "at world size 1) as the headline instead of the drop-in boundary")
    ap.add_argument("--stepwise", action="store_true",
                    help="N=1: the drop-in call with sweep_width=1 (ADI steps one at a time, as the reference)")
    ap.add_argument("--col-split", type=int, default=0,
                    help="N>1: column parts per shift (0 = automatic: 2 when the ranks would otherwise "
                         "hold fewer than 4 groups)")
    ap.add_argument("--streams", type=int, default=1, help="Python driver only: concurrent batches per rank")
    ap.add_argument("--nts", type=int, default=8, help="cfg4-dre: time steps of the backward sweep")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4-dre"] + sorted(WORKLOADS),
                    help="cfg2 (default): the metric's configuration, one Newton step to K.  cfg3 / cfg4 / cfg5: "
                         "the larger BASELINE.json configurations as fixed-work shift cycles (see WORKLOADS)")
    ap.add_argument("--also-baseline-config", action="store_true",
                    help="N>1: after the cfg2 headline also time one shift cycle of the BASELINE configuration "
                         "quoted for this GPU count (cfg3 @ 4, cfg4 @ 8) and add it as `baseline_config_for_n`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sorted-shifts", action="store_true",
                    help="cfg3-cycle / cfg4 / cfg5 / cfg4-dre: the shift list in ascending |p| (rounds 1-3) instead of interleaved")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N>1 on a one-GPU box: all ranks on device 0 with gloo (checks the code path only)")
    ap.add_argument("--cpu-full", action="store_true",
                    help="cpu_baseline: run the whole single-core step (all LUs, all panel solves, ~100 s)")
    ap.add_argument("--no-large-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="headline only (no second figures, no kernel rooflines)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # start the N ranks ourselves: a CHILD torchrun (never an exec), decided before torch or the HIP
        # library have been imported; rank 0's JSON line and the exit code are relayed
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "2")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        for ln in r.stdout.splitlines():          # ONE JSON line on stdout; library chatter (gloo) goes to stderr
            print(ln, file=sys.stdout if ln.startswith("{") else sys.stderr, flush=True)
        sys.exit(r.returncode)

    import torch
    import torch.distributed as dist
    from optconpy_amd import _lib, backend, problems as pb
    import sadptprj_riclyap_adi.proj_ric_utils as pru

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and args.rehearse_one_gpu:
        # developer rehearsal of the N > 1 code path on a one-GPU box: every rank on device 0, gloo collectives
        # (RCCL refuses two ranks on one device); the figure it prints is not a measurement
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local = 0
        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
        os.environ["LOCAL_RANK"] = "0"      # the drop-in backend picks its device from it
    elif world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0:
        log("note: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world))
    os.environ["RICADI_DEVICE"] = str(local)

    def bar(cx):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        cx.synchronize()

    if args.workload == "cfg4-dre":
        line = dre_workload(args, rank)
        if rank == 0:
            print(json.dumps(line), flush=True)
        backend.reset()
        if world > 1:
            dist.destroy_process_group()
        return
    if args.workload in WORKLOADS:
        res = cycle_workload(args.workload, world, rank, local, args.steps, args.warmup, args.col_split, bar,
                             python_driver=args.python_driver, with_cpu_baseline=not args.no_cpu_baseline,
                             interleave=not args.sorted_shifts)
        if rank == 0:
            line = {
                "metric": "ADI shift-solves/sec", "value": res["value"], "unit": "shift-solves/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
                "data": "synthetic", "config": res}
            if "cpu_baseline" in res:
                line["cpu_baseline"] = res.pop("cpu_baseline")
            print(json.dumps(line), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if args.workload == "cfg3":
        # BASELINE cfg3 as what it is: the steady-state Riccati run (cyl_wake_cont.py:34-50 -> optcont_main.py:488-506)
        # on the surrogate of SURVEY.md 8d -- the Newton step to K below at N = 75, nu = 0.15 / 40, 32 shifts
        args.N, args.nu, args.shifts = CFG3["N"], CFG3["nu"], CFG3["shifts"]

    xopts = _xopts()
    if xopts:
        backend.configure(**xopts)
    t0 = time.time()
    il = args.workload == "cfg3" and not args.sorted_shifts      # 32 shifts: interleaved so that sweeps of 16 are admissible
    pr, tb, trct, ms = build_inputs(args.N, args.nu, args.shifts, interleave=il)
    F = (-pr.A - pr.Nc).tocsr()
    MT = pr.M.T.tocsr()
    calA, calE = F.T.tocsr(), MT
    d = dict(pb.default_nwtn_adi_dict(), ms=ms, sweep_width=1 if args.stepwise else args.sweep_width)
    nb, mw = tb.shape[1], trct.shape[1]
    m = nb + mw
    n = pr.NV + pr.NP
    nnz_k = (calA + calE).nnz
    nnz_s = nnz_k + 2 * pr.J.nnz
    K_oracle = oracle_gain(args.N, args.nu, ms)

    # untimed: converge the Newton iteration once through the boundary; its compressed
    # iterate is the linearisation point Z_k of the timed Newton step
    conv = pru.proj_alg_ric_newtonadi(mmat=pr.M, amat=F, jmat=pr.J, bmat=tb, wmat=trct, nwtn_adi_dict=d)
    Zk = conv["zfac"]
    K_conv = -pru.get_mTzzTtb(MT, Zk, tb)
    ctx = backend.context_for(calA, calE, pr.J)          # the drop-in's context (operator resident)
    if rank == 0:
        log("setup %.1fs: n=%d nnz(S)=%d m=%d; converged Newton: %d steps, %d shift-solves; |Z_k| cols %d; "
            "K vs oracle fixture: %s"
            % (time.time() - t0, n, nnz_s, m, conv["nwtn_steps"], conv["shift_solves"], Zk.shape[1],
               "n/a" if K_oracle is None else "%.2e" % (np.linalg.norm(K_conv - K_oracle) / np.linalg.norm(K_oracle))))
    d1 = dict(d, nwtn_max_steps=1)

    # the step's input panels B~, C~^T, Z_k staged in HBM once, before any timed region (`value` is measured with
    # the inputs resident in HBM; the same step fed with host arrays is reported as value_pcie_inclusive)
    tb_d, trct_d, Zk_d = pru.to_device(tb), pru.to_device(trct), pru.to_device(Zk)

    def _dropin(bm, wm, z0):
        ctx.clear_cache()
        out = pru.proj_alg_ric_newtonadi(mmat=pr.M, amat=F, jmat=pr.J, bmat=bm, wmat=wm, z0=z0,
                                         nwtn_adi_dict=d1)
        K = -pru.get_mTzzTtb(MT, out["zfac"], bm)
        if out["gmres_nonconverged"]:
            raise RuntimeError("bench: %d timed shift-solves missed gmres_tol (worst %.2e)"
                               % (out["gmres_nonconverged"], out["gmres_worst_relres"]))
        return out["adi_steps"], out["gmres_iters"], K, out["shift_solves"]

    def dropin_step():
        """optcont_main.py:488-492,505 through the boundary; per-shift setup is part of the step.  Panels in HBM:
        the new factor stays there (DeviceFactor), the gain K (NV x 8) comes back to the host."""
        return _dropin(tb_d, trct_d, Zk_d)

    def dropin_step_host():
        """The same step with ndarray panels, as the reference's callers hand them over: uploads of B~, C~^T, Z_k
        and the download of the new factor inside the step."""
        return _dropin(tb, trct, Zk)

    # the Python sweep driver (the multi-GPU code path; at world size 1 a second figure)
    from optconpy_amd.shift_parallel import HipOps, lyap_adi_shift_parallel, plan_items
    from optconpy_amd import lin_alg_utils as glau
    sp = {}

    def sp_prepare():
        nstreams = max(1, args.streams)
        extra = []
        for _ in range(max(0, nstreams - 1)):
            cx = _lib.Context(local)
            cx.set_operator(calA, calE, pr.J)
            extra.append(cx)
        ops = HipOps(ctx, extra)
        G = max(1, min(args.sweep_width, 16, len(ms)))
        Kk = -K_conv                                         # K_k = E Z_k Z_k^T B
        Wp = glau.app_prj_via_sadpnt(amat=pr.M, jmat=pr.J, rhsv=trct, transposedprj=True)
        backend.context_for(calA, calE, pr.J)                # the projection re-set the operator (E only)
        sp.update(ops=ops, G=G, Kk=Kk, rhs=ops.to_panel(np.hstack([Wp, Kk])), tbd=ops.to_panel(tb),
                  extra=extra, parts=plan_items(G, world, args.col_split))

    def sp_step():
        ops = sp["ops"]
        ops.clear_cache()
        ops.set_lowrank(sp["Kk"], tb)
        ops.gmres_iters = ops.shift_solves = ops.nonconverged = 0
        try:
            blocks, info = lyap_adi_shift_parallel(ops, ms, sp["rhs"], adi_max_steps=200,
                                                   adi_newZ_reltol=1e-8, width=sp["G"],
                                                   col_parts=sp["parts"])
        finally:
            ops.set_lowrank(None, None)
        if info["gmres_nonconverged"]:
            raise RuntimeError("bench: %d timed shift-solves missed gmres_tol (worst %.2e)"
                               % (info["gmres_nonconverged"], info["gmres_worst_relres"]))
        Z = torch.cat(blocks, dim=1).contiguous()
        Kt = ops.gain(-1.0, Z, sp["tbd"])                    # gain on the replicated factor
        return info["adi_steps"], ops.gmres_iters, Kt.cpu().numpy(), ops.shift_solves

    use_sp = args.python_driver        # N > 1 runs the SAME boundary step as N = 1 (the library shards the sweeps)
    if use_sp:
        sp_prepare()
    one_step = sp_step if use_sp else dropin_step

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    step_ms = []

    def timed(step, nsteps, nwarm):
        for _ in range(nwarm):
            step()
        barrier()
        t0 = time.perf_counter()
        units = iters = solves = 0
        K = None
        step_ms.clear()
        for _ in range(nsteps):
            ts = time.perf_counter()
            u, it, K, ls = step()
            units += u
            iters += it
            solves += ls
            step_ms.append(round(1e3 * (time.perf_counter() - ts), 1))
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return units, iters, solves, K, el

    units, iters, local_solves, K, elapsed = timed(one_step, args.steps, args.warmup)
    main_step_ms = list(step_ms)
    k_conv = float(np.linalg.norm(K - K_conv) / np.linalg.norm(K_conv))
    k_orc = None if K_oracle is None else float(np.linalg.norm(K - K_oracle) / np.linalg.norm(K_oracle))

    if rank == 0:
        basis = "FP64" if os.environ.get("RICADI_BASIS64") else "FP32" if os.environ.get("RICADI_BASIS32") else "FP16"
        prec = "FP64" if os.environ.get("RICADI_PRECOND64") else "FP32"
        if use_sp:
            par = ("shift-parallel ADI on %d GPU(s) (Python sweep driver, Lyapunov sweeps + gain only): %d shifts/sweep x "
                   "%d column part(s) = %d work items dealt to the ranks, one batched lockstep solve per rank and "
                   "sweep, 1 all-gather/sweep" % (world, sp["G"], sp["parts"], sp["G"] * sp["parts"]))
        else:
            par = ("drop-in boundary (sadptprj_riclyap_adi.proj_ric_utils.proj_alg_ric_newtonadi + get_mTzzTtb; input "
                   "panels B~, C~^T, Z_k resident in HBM, the new factor stays there, K returned to the host), "
                   "C++ Newton-ADI, %s, %s"
                   % ("ADI steps one at a time" if args.stepwise
                      else "sweeps of %d shifts in one batched lockstep solve" % args.sweep_width,
                      "1 GPU" if world == 1 else
                      "%d GPUs: the same full step on every rank count -- sweeps sharded by shift inside the library "
                      "(owners %s), one all-gather of the solution panels per sweep over %s, everything else replicated"
                      % (world, list(map(int, _lib.host_deal(ms, world))),
                         "gloo (one-GPU rehearsal)" if args.rehearse_one_gpu else "RCCL")))
        out = {
            "metric": "ADI shift-solves/sec (wall-clock to feedback K in ms_per_step)",
            "value": round(units / elapsed, 3),
            "unit": "shift-solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2),
            "ms_of_each_step": main_step_ms,
            "higher_is_better": True,
            "scaling": "strong",          # the same problem at every N
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s: driven cavity N=%d (n=%d, nnz(S)=%d), nu=%g, %d log-spaced ADI shifts%s, "
                            "1 Newton step from the converged iterate (closed-loop Lyapunov ADI to "
                            "adi_newZ_reltol=1e-8, update norm, recompression) + gain K; panel m=%d; "
                            "GMRES tol 1e-10; per-shift setup inside the step"
                            % ("cfg3 (steady-state Riccati, cylinder-wake surrogate)" if args.workload == "cfg3"
                               else "cfg2" if args.N == 58 else "custom", args.N, n, nnz_s, args.nu, len(ms),
                               " (interleaved order)" if il else "", m),
                "preconditioner_levels": ctx.setup_info()["levels"],
                "shift_solves_per_step": units // args.steps,
                "gmres_iters_per_shift_solve": round(iters / max(local_solves, 1), 1),
                "parallelism": par,
                "K_rel_diff_vs_oracle": k_orc,
                "K_rel_diff_vs_converged": k_conv,
                "storage": "arithmetic and all residual checks FP64; Krylov basis stored in %s, "
                           "Z_j = P^-1 v_j in FP32, the operator's output w inside the iteration in %s, "
                           "preconditioner: coarse inverse in %s, block operands of the sweeps in %s, the velocity part "
                           "between its sweeps in %s (RICADI_BASIS64=1 RICADI_PRECOND64=1: FP64 storage of basis, w and "
                           "preconditioner)"
                           % (basis, "FP32" if ctx.setup_info().get("fp32_operator_output", 0) == 1 else "FP64", prec,
                              "BF16" if prec == "FP32" and os.environ.get("RICADI_BLOCKS16", "1") != "0"
                              and os.environ.get("RICADI_MID32", "1") != "0" else prec,
                              "FP32" if prec == "FP32" and os.environ.get("RICADI_MID32", "1") != "0" else "FP64"),
            },
        }
        if k_orc is not None and k_orc > 1e-6:
            out["config"]["PARITY_VIOLATION"] = "K differs from the oracle fixture by %.2e > 1e-6" % k_orc

    extras = world == 1 and not args.no_extras
    if extras:
        # --- second figures on the same workload (untimed by the driver) -------------------
        try:
            if use_sp:
                u2, _, _, K2, el2 = timed(dropin_step, 2, 1)
                out["value_dropin_boundary"] = round(u2 / el2, 3)
            else:
                u2, _, _, K2, el2 = timed(dropin_step_host, 2, 1)
                out["value_pcie_inclusive"] = round(u2 / el2, 3)
                out["ms_per_step_pcie_inclusive"] = round(1e3 * el2 / 2, 2)
                out["pcie_inclusive_note"] = ("the same step with ndarray panels: B~, C~^T, Z_k (%.0f MB) uploaded and "
                                              "the new factor downloaded inside the step"
                                              % ((tb.nbytes + trct.nbytes + Zk.nbytes) / 1e6))
                out["pcie_inclusive_K_rel_diff_vs_oracle"] = (
                    None if K_oracle is None else float(np.linalg.norm(K2 - K_oracle) / np.linalg.norm(K_oracle)))
                sp_prepare()
                u2, _, _, K2, el2 = timed(sp_step, 2, 1)
                out["value_python_sweep_driver"] = round(u2 / el2, 3)
                out["python_sweep_driver_K_rel_diff_vs_oracle"] = (
                    None if K_oracle is None else float(np.linalg.norm(K2 - K_oracle) / np.linalg.norm(K_oracle)))
        except Exception as e:
            out["second_figure_error"] = str(e)
        if args.workload == "cfg3":
            # continuation in Re (optcont_main.py:471-486, cyl_wake_cont.py:37-45): the Newton iteration at nu started
            # from the iterate of a run at twice the viscosity, to convergence, through the boundary
            try:
                pr2, tb2, trct2, _ = build_inputs(args.N, 2.0 * args.nu, args.shifts, interleave=il)
                F2 = (-pr2.A - pr2.Nc).tocsr()
                low = pru.proj_alg_ric_newtonadi(mmat=pr2.M, amat=F2, jmat=pr2.J, bmat=tb2, wmat=trct2, nwtn_adi_dict=d)
                z0 = pru.compress_Zsvd(low["zfac"], thresh=1e-8, k=400)
                backend.context_for(calA, calE, pr.J)
                tz = time.perf_counter()
                oz = pru.proj_alg_ric_newtonadi(mmat=pr.M, amat=F, jmat=pr.J, bmat=tb, wmat=trct, z0=z0, nwtn_adi_dict=d)
                Kz = -pru.get_mTzzTtb(MT, oz["zfac"], tb)
                tz = time.perf_counter() - tz
                out["continuation_from_lower_re"] = dict(
                    z0="converged iterate at nu = %g (twice the viscosity), compressed to %d columns" % (2.0 * args.nu, z0.shape[1]),
                    newton_steps=oz["nwtn_steps"], shift_solves=oz["shift_solves"], seconds_to_K=round(tz, 3),
                    newton_steps_from_zero=conv["nwtn_steps"],
                    K_rel_diff_vs_oracle=None if K_oracle is None else
                    float(np.linalg.norm(Kz - K_oracle) / np.linalg.norm(K_oracle)))
                ctx = backend.context_for(calA, calE, pr.J)
            except Exception as e:
                out["continuation_from_lower_re"] = {"error": str(e)}
        # --- kernel rooflines on the launches of this workload ---------------------------------
        G = max(1, min(args.sweep_width, 16, len(ms)))
        gsh = [float(p) for p in ms[:G]]
        out["roofline"] = spmm_roofline(ctx, nnz_k, pr.J.nnz, n, m, gsh)
        try:
            out["roofline_kernels"] = kernel_rooflines(ctx, gsh, m)
        except Exception as e:
            out["roofline_kernels"] = {"error": str(e)}
        out["roofline_gram_mfma"] = gram_mfma(ctx, pr.NV, 512)
        try:
            out["roofline_tsqr_mfma"] = tsqr_mfma(ctx, pr.NV, 456)
        except Exception as e:
            out["roofline_tsqr_mfma"] = {"error": str(e)}
        # --- FP64 storage of basis + preconditioner: the like-for-like number -----------------
        try:
            saved = {k: os.environ.get(k) for k in ("RICADI_BASIS64", "RICADI_PRECOND64")}
            os.environ["RICADI_BASIS64"] = "1"
            os.environ["RICADI_PRECOND64"] = "1"
            backend.reset()
            ctx = backend.context_for(calA, calE, pr.J)
            u3, it3, _, K3, el3 = timed(dropin_step, 2, 1)
            out["value_fp64_storage"] = round(u3 / el3, 3)
            out["fp64_storage_K_rel_diff_vs_oracle"] = (
                None if K_oracle is None else float(np.linalg.norm(K3 - K_oracle) / np.linalg.norm(K_oracle)))
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            backend.reset()
            ctx = backend.context_for(calA, calE, pr.J)
        except Exception as e:
            out["value_fp64_storage"] = {"error": str(e)}
        if not args.no_large_roofline:
            # the HBM-resident instance (BASELINE cfg5 pattern, n ~ 5e5)
            try:
                prl = pb.ricc_problem(236, 0.05, with_convection=True)
                cl = _lib.Context(local)
                cAl, cEl = (-prl.A - prl.Nc).T.tocsr(), prl.M.T.tocsr()
                cl.set_operator(cAl, cEl, prl.J)
                out["roofline_cfg5"] = spmm_roofline(cl, (cAl + cEl).nnz, prl.J.nnz, prl.NV + prl.NP, 16, gsh,
                                                     reps=50)
                try:
                    out["roofline_kernels_cfg5"] = kernel_rooflines(cl, gsh, 16, reps=20)
                except Exception as e:
                    out["roofline_kernels_cfg5"] = {"error": str(e)}
                cl.close()
            except Exception as e:                   # never lose the headline line
                out["roofline_cfg5"] = {"error": str(e)}
    elif rank == 0 and world == 1:
        G = max(1, min(args.sweep_width, 16, len(ms)))
        out["roofline"] = spmm_roofline(ctx, nnz_k, pr.J.nnz, n, m, [float(p) for p in ms[:G]])
    elif rank == 0:
        # N > 1: the launch a rank issues holds its own shifts of the sweep
        own = [float(p) for p, o in zip(ms, _lib.host_deal(ms, world)) if o == 0][:16]
        out["roofline"] = spmm_roofline(ctx, nnz_k, pr.J.nnz, n, m, own or [float(ms[0])])
    if world > 1 and args.also_baseline_config:
        cfg_for_n = {4: "cfg3-cycle", 8: "cfg4"}.get(world)
        if cfg_for_n:
            try:
                extra = cycle_workload(cfg_for_n, world, rank, local, 1, 0, args.col_split, bar)
                if rank == 0:
                    out["baseline_config_for_n"] = extra
            except Exception as e:
                if rank == 0:
                    out["baseline_config_for_n"] = {"error": str(e)}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pr, ms, m, units // args.steps, full=args.cpu_full)
        print(json.dumps(out), flush=True)
    backend.reset()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
