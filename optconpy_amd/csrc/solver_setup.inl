// solver_setup.inl -- per-shift setup: workspaces, coarse inverses (block Gauss-Jordan / pivoted rocSOLVER), block inverses, Schur blocks.
// Part of ricadi_solver.hip (one translation unit; included there in order).


// Width of the column groups a wide panel is solved in (0: the panel stays whole); see gmres_core_any.
static int wide_split_width(const ricadi_ctx* c, int m) {
  static const int off = getenv("RICADI_WIDE_SPLIT") && atoi(getenv("RICADI_WIDE_SPLIT")) == 0 ? 1 : 0;
  (void)c;
  return (!off && m > 32) ? 16 : 0;
}

// Workspace for batches of up to `groups` panels of width m (group-major: every
// buffer holds one slab per group; basis is vector-major, i.e. Krylov vector j of
// all groups is contiguous).
// `extra` columns per group are reserved on top of m (default: the low-rank width, for
// the augmented Sherman-Morrison-Woodbury solves) so that a nested, wider solve never
// reallocates buffers the caller has already filled.
static void ensure_work(ricadi_ctx* c, int m, int groups = 1, int extra = -1) {
  const int restart = c->opts.gmres_restart;
  if (extra < 0) extra = std::max(c->q, 0);
  if (c->child) {
    c->child->opts.gmres_restart = std::min(c->opts.gmres_restart, 4);   // its Krylov buffers are not used
    ensure_work(c->child.get(), m, groups, extra);
  }
  // every buffer scales with the total number of columns (m + extra) * groups
  int want = (m + extra) * groups;
  // a wide panel is solved as chunks of up to RICADI_MAX_GROUPS sixteen-column groups: the buffers must hold a
  // full chunk already NOW -- the caller's right-hand side lives in them (c->bvec) when the solve starts
  if (wide_split_width(c, m + extra)) want = std::max(want, 16 * RICADI_MAX_GROUPS);
  if (want <= c->wcols && restart == c->wrestart) return;
  const size_t gm = (size_t)std::max(want, c->wcols);
  const size_t nm = (size_t)c->n * gm;
  // Krylov basis: stored in FP16 by default (FP32 with RICADI_BASIS32=1, FP64 with
  // RICADI_BASIS64=1); ALL arithmetic stays FP64 -- the three passes over the basis per
  // iteration are the largest share of the HBM traffic.  The current vector is also
  // kept in FP64 (vcur, holding the same rounded values) for the operator /
  // preconditioner application, so the Arnoldi relation holds exactly for the stored
  // vectors; what the storage precision limits is the residual reduction one restart
  // cycle can deliver (~1e-3 for FP16, cycles gain ~1e-2), and every cycle starts from
  // the true FP64 residual.  Unit vectors of dimension n have entries ~ n^-1/2: FP16
  // (normal range from 6e-5) is used up to n = 2^21, FP32 beyond.
  c->basis32 = getenv("RICADI_BASIS64") == nullptr;
  c->basis16 = c->basis32 && getenv("RICADI_BASIS32") == nullptr && c->n <= (1 << 21);
  if (c->basis32) {
    c->basisf.alloc((size_t)(restart + 1) * nm);
    c->vcur.alloc(nm);
    c->basis.release();
  } else {
    c->basis.alloc((size_t)(restart + 1) * nm);
    c->basisf.release();
  }
  c->flex = true;      // flexible GMRES: Z_j = P^-1 v_j kept (FP32), x += Z y at the cycle end
  if (c->flex) c->zbasisf.alloc((size_t)restart * nm);
  else c->zbasisf.release();
  c->wv.alloc(nm);
  c->wv32.alloc(nm);
  c->zv.alloc(nm);
  c->r2.alloc(nm);
  c->xs.alloc(nm);
  c->bvec.alloc(nm);
  c->pw1.alloc(nm);
  c->pw2.alloc(nm);
  c->tp.alloc((size_t)std::max(c->np, 1) * gm);
  c->rc.alloc((size_t)std::max(c->kc, 1) * gm);
  c->ec.alloc((size_t)std::max(c->kc, 1) * gm);
  c->partial.alloc((size_t)dots_num_blocks(c->n) * (restart + 2) * gm);
  c->h1.alloc((size_t)(restart + 2) * gm);
  c->h2.alloc((size_t)2 * (restart + 2) * gm);      // two buffers (atomic dot passes alternate between them)
  c->H.alloc(gm * (restart + 1) * restart);
  c->cs.alloc(gm * restart);
  c->sn.alloc(gm * restart);
  c->g.alloc(gm * (restart + 1));
  c->scale.alloc(gm);
  c->resid.alloc(2 * gm);          // two buffers (the fused update + Hessenberg launch alternates between them)
  c->yv.alloc((size_t)restart * gm);
  c->bnorm2.alloc(gm);
  c->nrm2.alloc(gm);
  c->lrc.alloc((size_t)64 * gm + 64);
  if (!c->h_resid) {
    HIPCHK(hipHostMalloc((void**)&c->h_resid,
                         sizeof(double) * 4 * RICADI_MAX_M * RICADI_MAX_GROUPS));
    for (int i = 0; i < 2; ++i) HIPCHK(hipEventCreateWithFlags(&c->ev_res[i], hipEventDisableTiming));
  }
  c->wcols = (int)gm;
  c->wrestart = restart;
}

// ---- per-shift setup ---------------------------------------------------------
template <class T>
static void stable_alloc(DArr<T>& a, size_t n) {
  if (a.n != n) a.alloc(n);
}

// In-place inverses of nb (<= RICADI_MAX_GROUPS) dense k x k matrices (row-major, device pointers in hmats) by
// block Gauss-Jordan elimination without pivoting: per 128-row block three small kernels and two batched
// rocBLAS GEMMs (ricadi_kernels.hip).  Returns false if a diagonal block had a vanishing pivot (the matrices
// are garbage then; the caller assembles them again and takes the pivoted rocSOLVER route).
static bool gj_invert_batched(ricadi_ctx* c, double* const* hmats, int nb, int k) {
  hipStream_t st = c->st;
  const int NB = gj_block();
  const size_t pan = (size_t)k * NB;
  c->gj_cb.ensure(pan * nb);
  c->gj_rp.ensure(pan * nb);
  c->gj_rb.ensure(pan * nb);
  c->gj_d.ensure((size_t)NB * NB * nb);
  std::vector<double*> hp((size_t)5 * nb);
  for (int i = 0; i < nb; ++i) {
    hp[i] = hmats[i];
    hp[nb + i] = c->gj_cb.p + pan * i;
    hp[2 * nb + i] = c->gj_rp.p + pan * i;
    hp[3 * nb + i] = c->gj_rb.p + pan * i;
    hp[4 * nb + i] = c->gj_d.p + (size_t)NB * NB * i;
  }
  c->gj_ptrs.ensure((size_t)5 * nb);
  HIPCHK(hipMemcpyAsync(c->gj_ptrs.p, hp.data(), sizeof(double*) * hp.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(c->flag.p + 2, 0, sizeof(int), st));
  HIPCHK(hipStreamSynchronize(st));   // hp is a stack object
  double* const* dA = c->gj_ptrs.p;
  double* const* dCb = dA + nb;
  double* const* dRp = dA + 2 * nb;
  double* const* dRb = dA + 3 * nb;
  double* const* dD = dA + 4 * nb;
  const double one = 1.0, zero = 0.0, mone = -1.0;
  for (int k0 = 0; k0 < k; k0 += NB) {
    const int nbe = std::min(NB, k - k0);
    launch_gj_prep(st, nb, hmats, k, k0, nbe, c->gj_cb.p, c->gj_rp.p, c->gj_d.p);
    launch_gj_diag(st, nb, c->gj_d.p, nbe, c->flag.p + 2);
    // row-major Rb = D^-1 Rp  ==  column-major Rb^T = Rp^T (D^-1)^T
    RBCHK(rocblas_dgemm_batched(c->rb, rocblas_operation_none, rocblas_operation_none, k, nbe, nbe, &one,
                                (const double* const*)dRp, k, (const double* const*)dD, NB, &zero, dRb, k, nb));
    // row-major A -= Cb Rb  ==  column-major A^T -= Rb^T Cb^T
    RBCHK(rocblas_dgemm_batched(c->rb, rocblas_operation_none, rocblas_operation_none, k, k, nbe, &mone,
                                (const double* const*)dRb, k, (const double* const*)dCb, NB, &one, dA, k, nb));
    launch_gj_rows(st, nb, hmats, k, k0, nbe, c->gj_rb.p);
  }
  int flag = 0;
  HIPCHK(hipMemcpyAsync(&flag, c->flag.p + 2, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return flag == 0;
}

// In-place inverses of nb dense k x k matrices (row major, device pointers in hp): the coarse matrices of a setup.
// Route 0: block Gauss-Jordan WITHOUT pivoting on batched GEMMs (gj_invert_batched) -- with the velocity
// aggregates ordered before the pressure aggregates that is block elimination of the coarse saddle matrix: the
// velocity block has a definite symmetric part for ADI shifts (and is s.p.d. for the projection), the Schur
// complement -B Av^-1 B^T inherits it.  A pivot that vanishes relative to its block's scale sends ALL matrices of
// the call through route 1: rocSOLVER's getrf / getri with partial pivoting (its unpivoted routines, the step in
// between until round 3, only notice an EXACTLY zero pivot -- a pivot of 1e-14 of the block's scale passed and left
// a garbage inverse).  `reassemble` restores the matrices the first route has overwritten.  info (nb entries):
// rocSOLVER's status.  Returns the route.
template <class F>
static int invert_dense_batch(ricadi_ctx* c, const std::vector<double*>& hp, int k, std::vector<int>& info,
                              F&& reassemble) {
  hipStream_t st = c->st;
  const int nb = (int)hp.size();
  bool done = true;
  for (int i0 = 0; i0 < nb && done; i0 += gj_max_batch())
    done = gj_invert_batched(c, hp.data() + i0, std::min(gj_max_batch(), nb - i0), k);
  if (done) {
    std::fill(info.begin(), info.end(), 0);
    return 0;
  }
  reassemble();
  // row-major E == column-major E^T; inv(E^T) column-major == inv(E) row-major
  c->ipiv.ensure((size_t)k * nb);
  c->info.ensure(nb);
  c->eptrs.ensure(nb);
  HIPCHK(hipMemcpyAsync(c->eptrs.p, hp.data(), sizeof(double*) * nb, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));   // hp may be a stack object of the caller
  RBCHK(rocsolver_dgetrf_batched(c->rb, k, k, c->eptrs.p, k, c->ipiv.p, k, c->info.p, nb));
  RBCHK(rocsolver_dgetri_batched(c->rb, k, c->eptrs.p, k, c->ipiv.p, k, c->info.p, nb));
  HIPCHK(hipMemcpyAsync(info.data(), c->info.p, sizeof(int) * nb, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 1;
}

// Per-shift data for the given (alpha, beta) pairs; whatever is missing is built for
// all of them together: the element-wise / block kernels per shift, the dense coarse
// inverses in ONE batched rocSOLVER factorisation + inversion (its many small
// panel kernels then serve all shifts of a sweep per launch instead of one).
static void get_shifts(ricadi_ctx* c, const double* alphas, const double* betas, int ng,
                       ShiftData** out) {
  hipStream_t st = c->st;
  std::vector<ShiftData*> todo;
  for (int g = 0; g < ng; ++g) {
    auto key = std::make_pair(alphas[g], betas[g]);
    auto it = c->cache.find(key);
    if (it == c->cache.end()) it = c->cache.emplace(key, std::unique_ptr<ShiftData>(new ShiftData)).first;
    ShiftData* sd = it->second.get();
    out[g] = sd;
    if (sd->valid || std::find(todo.begin(), todo.end(), sd) != todo.end()) continue;
    sd->alpha = alphas[g];
    sd->beta = betas[g];
    sd->smw_epoch = -1;
    for (auto& r : sd->rec) r->serial = -1;     // stale, but the buffers stay (a hipFree / hipMalloc pair per panel
                                                // cost 13 ms per setup of 17 shifts)
    todo.push_back(sd);
  }
  if (todo.empty()) return;
  Tick tks;
  double tph[6] = {0, 0, 0, 0, 0, 0};
  auto lapS = [&](int i) {
    if (c->timing) {
      (void)hipStreamSynchronize(st);
      tph[i] += tks.lap();
    }
  };
  HIPCHK(hipMemsetAsync(c->flag.p, 0, sizeof(int), st));
  const size_t bsz = (size_t)c->bs * c->bs;
  const int k = c->kc;
  const int kd = c->child ? 0 : c->kc;   // size of the dense coarse inverse (none with a child level)
  if (c->child) {
    std::vector<double> al(todo.size()), be(todo.size());
    std::vector<ShiftData*> subs(todo.size(), nullptr);
    for (size_t i = 0; i < todo.size(); ++i) {
      al[i] = todo[i]->alpha;
      be[i] = todo[i]->beta;
    }
    get_shifts(c->child.get(), al.data(), be.data(), (int)todo.size(), subs.data());
    for (size_t i = 0; i < todo.size(); ++i) todo[i]->sub = subs[i];
  }
  lapS(0);
  for (ShiftData* sd : todo) {
    const double alpha = sd->alpha, beta = sd->beta;
    stable_alloc(sd->sval, c->snnz);
    launch_assemble_shift(st, (int)c->snnz, c->srcA.p, c->srcE.p, c->srcJ.p, alpha, beta,
                          sd->sval.p);
    if (c->sb_ok) {
      stable_alloc(sd->svalb, c->snnz);
      launch_gather_vals(st, (int)c->snnz, c->sb_perm.p, sd->sval.p, sd->svalb.p);
    }
    stable_alloc(sd->bvinv, (size_t)c->nbv * bsz);
    launch_block_combine(st, (size_t)c->nbv * bsz, c->bvA.p, c->bvE.p, alpha, beta, sd->bvinv.p);
    if (c->nbp > 0) stable_alloc(sd->bpinv, (size_t)c->nbp * bsz);
    if (c->gt_ok) stable_alloc(sd->gtm, (size_t)c->nbv * c->bs * c->gt_ks);
    if (c->ady_ok && k > 0) stable_alloc(sd->adym, (size_t)c->nbv * c->bs * c->ady_ks);
    if (kd > 0) {
      stable_alloc(sd->einv, (size_t)k * k);
      launch_combine3(st, (size_t)k * k, c->E0.p, c->EM.p, c->EJ.p, alpha, beta, sd->einv.p);
    }
    if (k > 0) {
      stable_alloc(sd->syval, c->synnz);
      launch_assemble_shift(st, (int)c->synnz, c->sy_A.p, c->sy_E.p, c->sy_J.p, alpha, beta,
                            sd->syval.p);
      if (c->syb_ok) {
        stable_alloc(sd->syvalb, c->synnz);
        launch_gather_vals(st, (int)c->synnz, c->syb_perm.p, sd->syval.p, sd->syvalb.p);
      }
    }
  }
  lapS(1);
  // block inversions and Schur blocks: one launch each for all shifts (<= 16 per call)
  for (size_t t0 = 0; t0 < todo.size(); t0 += RICADI_MAX_GROUPS) {
    const int cnt = (int)std::min<size_t>(RICADI_MAX_GROUPS, todo.size() - t0);
    GroupPtrs pv = same_ptr((const double*)nullptr), pp = pv;
    for (int i = 0; i < cnt; ++i) {
      pv.p[i] = todo[t0 + i]->bvinv.p;
      pp.p[i] = todo[t0 + i]->bpinv.p;
    }
    launch_block_invert(st, cnt, c->nbv, c->bs, c->bv_ptr.p, pv, c->flag.p);
    if (c->gt_ok) {
      GroupPtrs pg = same_ptr((const double*)nullptr);
      for (int i = 0; i < cnt; ++i) pg.p[i] = todo[t0 + i]->gtm.p;
      launch_gt_blocks(st, cnt, c->nbv, c->bs, c->gt_ks, c->gt_jtd.p, pv, pg);
    }
    if (c->ady_ok && k > 0) {
      GroupPtrs pa_ = same_ptr((const double*)nullptr);
      double al[RICADI_MAX_GROUPS], be[RICADI_MAX_GROUPS];
      for (int i = 0; i < cnt; ++i) {
        pa_.p[i] = todo[t0 + i]->adym.p;
        al[i] = todo[t0 + i]->alpha;
        be[i] = todo[t0 + i]->beta;
      }
      launch_ady_blocks(st, cnt, c->nbv, c->bs, c->ady_ks, c->cy_dA.p, c->cy_dE.p, c->cy_dJ.p,
                        c->sa ? c->cy_dT.p : nullptr, al, be, pv, pa_);
    }
    if (c->nbp > 0) {
      launch_schur_blocks_bj(st, cnt, c->nbp, c->bs, c->bp_ptr.p, c->jd_ptr.p, c->jd_vblk.p,
                             c->jd_val.p, pv, pp);
      launch_block_invert(st, cnt, c->nbp, c->bs, c->bp_ptr.p, pp, c->flag.p);
    }
  }
  lapS(2);
  const int nb = (int)todo.size();
  std::vector<int> info(nb, 0);
  if (kd > 0) {
    std::vector<double*> hp(nb);
    for (int i = 0; i < nb; ++i) hp[i] = todo[i]->einv.p;
    c->coarse_route = invert_dense_batch(c, hp, k, info, [&] {
      for (ShiftData* sd : todo)
        launch_combine3(st, (size_t)k * k, c->E0.p, c->EM.p, c->EJ.p, sd->alpha, sd->beta, sd->einv.p);
    });
  }
  lapS(3);
  int flag = 0;
  HIPCHK(hipMemcpyAsync(&flag, c->flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for (int i = 0; i < nb; ++i)
    if (info[i] != 0)
      throw HipError{"coarse matrix singular (getrf/getri info " + std::to_string(info[i]) + ")"};
  if (flag) throw HipError{"singular block-Jacobi block"};
  if (c->precond32) {
    const int bs2 = c->bs * c->bs;
    for (ShiftData* sd : todo) {
      if (sd->bvinvf.n != sd->bvinv.n) sd->bvinvf.alloc(sd->bvinv.n);
      launch_to_f32(st, c->nbv, bs2, sd->bvinv.p, bs2, sd->bvinvf.p, bs2);
      if (c->nbp > 0) {
        if (sd->bpinvf.n != sd->bpinv.n) sd->bpinvf.alloc(sd->bpinv.n);
        launch_to_f32(st, c->nbp, bs2, sd->bpinv.p, bs2, sd->bpinvf.p, bs2);
      }
      if (c->gt_ok) {
        const int gsz = c->bs * c->gt_ks;
        if (sd->gtmf.n != sd->gtm.n) sd->gtmf.alloc(sd->gtm.n);
        launch_to_f32(st, c->nbv, gsz, sd->gtm.p, gsz, sd->gtmf.p, gsz);
      }
      if (c->ady_ok && k > 0) {
        const int gsz = c->bs * c->ady_ks;
        if (sd->adymf.n != sd->adym.n) sd->adymf.alloc(sd->adym.n);
        launch_to_f32(st, c->nbv, gsz, sd->adym.p, gsz, sd->adymf.p, gsz);
      }
      if (kd > 0) {
        const size_t kp = (size_t)(k + 15) / 16;
        if (sd->einvf.n != kp * kp * 256) sd->einvf.alloc(kp * kp * 256);
        launch_to_f32_tiled(st, k, sd->einv.p, sd->einvf.p);
      }
      if (c->blocks16 && c->sw_stride > 0 && c->gt_ok && c->ady_ok && k > 0 && c->nbp > 0) {
        // BF16 copies for the record-driven sweeps (all four or none: the cycle switches as a whole)
        if (sd->bvinvh.n != sd->bvinv.n) sd->bvinvh.alloc(sd->bvinv.n);
        if (sd->bpinvh.n != sd->bpinv.n) sd->bpinvh.alloc(sd->bpinv.n);
        if (sd->gtmh.n != sd->gtm.n) sd->gtmh.alloc(sd->gtm.n);
        if (sd->adymh.n != sd->adym.n) sd->adymh.alloc(sd->adym.n);
        launch_to_bf16(st, sd->bvinv.n, sd->bvinv.p, sd->bvinvh.p);
        launch_to_bf16(st, sd->bpinv.n, sd->bpinv.p, sd->bpinvh.p);
        launch_to_bf16(st, sd->gtm.n, sd->gtm.p, sd->gtmh.p);
        launch_to_bf16(st, sd->adym.n, sd->adym.p, sd->adymh.p);
      }
    }
    HIPCHK(hipStreamSynchronize(st));
  }
  lapS(4);
  if (c->timing && !c->borrowed)
    fprintf(stderr, "[ricadi timing] setup of %d shifts: child %.1f ms, per-shift assembly %.1f, block inverses + Schur blocks %.1f, coarse inverses %.1f, FP32 copies %.1f\n",
            (int)todo.size(), 1e3 * tph[0], 1e3 * tph[1], 1e3 * tph[2], 1e3 * tph[3], 1e3 * tph[4]);
  for (ShiftData* sd : todo) sd->valid = true;
}

static ShiftData* get_shift(ricadi_ctx* c, double alpha, double beta) {
  ShiftData* sd = nullptr;
  get_shifts(c, &alpha, &beta, 1, &sd);
  return sd;
}

