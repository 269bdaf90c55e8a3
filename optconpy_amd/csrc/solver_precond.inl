// solver_precond.inl -- batches of panels; the saddle operator and the multilevel preconditioner on device panels.
// Part of ricadi_solver.hip (one translation unit; included there in order).

// ---- batches -----------------------------------------------------------------------
// The shifts of one batched solve: per group id the shift-dependent operands, and
// the table of groups a launch works on (ricadi_internal.h).  All workspace
// buffers are group-major with the strides below.
struct Batch {
  int G = 0;                 // groups in the solve (ids 0 .. G-1)
  int m = 0;                 // panel width of every group
  GroupTab tab;              // groups the next launches act on
  double alpha[RICADI_MAX_GROUPS], beta[RICADI_MAX_GROUPS];   // shift of every group id
  GroupPtrs sval, svalb, syval, syvalb, bvinv, bpinv, einv;
  GroupPtrsF bvinvf, bpinvf, einvf;
  GroupPtrs gtm, adym;
  GroupPtrsF gtmf, adymf;
  GroupPtrsH bvinvh, bpinvh, gtmh, adymh;     // BF16 copies (null where a shift has none)
  bool blocks16 = false;                       // every group of the batch has them
  size_t gs = 0, gsp = 0, gsc = 0, gsq = 0;   // strides: n*m, np*m, kc*m, q*m
  std::shared_ptr<Batch> sub;                 // the same groups on the child level

  void all() {
    tab.ng = G;
    for (int g = 0; g < G; ++g) tab.gid[g] = g;
  }
  void only(int g) {
    tab.ng = 1;
    tab.gid[0] = g;
  }
  void set(const std::vector<int>& ids) {
    tab.ng = (int)ids.size();
    for (int i = 0; i < tab.ng; ++i) tab.gid[i] = ids[i];
  }
};

static Batch make_batch(ricadi_ctx* c, ShiftData* const* sds, int G, int m) {
  Batch bt;
  bt.G = G;
  bt.m = m;
  bt.tab = GroupTab{};
  bt.sval = bt.svalb = bt.syval = bt.syvalb = bt.bvinv = bt.bpinv = bt.einv = same_ptr((const double*)nullptr);
  bt.bvinvf = bt.bpinvf = bt.einvf = same_ptr((const float*)nullptr);
  bt.gtm = bt.adym = same_ptr((const double*)nullptr);
  bt.gtmf = bt.adymf = same_ptr((const float*)nullptr);
  bt.bvinvh = bt.bpinvh = bt.gtmh = bt.adymh = same_ptr((const uint16_t*)nullptr);
  bt.blocks16 = c->blocks16 && G > 0;
  for (int g = 0; g < RICADI_MAX_GROUPS; ++g) bt.alpha[g] = bt.beta[g] = 0.0;
  for (int g = 0; g < G; ++g) {
    bt.alpha[g] = sds[g]->alpha;
    bt.beta[g] = sds[g]->beta;
    bt.bvinvf.p[g] = sds[g]->bvinvf.p;
    bt.bpinvf.p[g] = sds[g]->bpinvf.p;
    bt.einvf.p[g] = sds[g]->einvf.p;
    bt.bvinvh.p[g] = sds[g]->bvinvh.p;
    bt.bpinvh.p[g] = sds[g]->bpinvh.p;
    bt.gtmh.p[g] = sds[g]->gtmh.p;
    bt.adymh.p[g] = sds[g]->adymh.p;
    if (!sds[g]->bvinvh.p || !sds[g]->bpinvh.p || !sds[g]->gtmh.p || !sds[g]->adymh.p) bt.blocks16 = false;
    bt.gtm.p[g] = sds[g]->gtm.p;
    bt.gtmf.p[g] = sds[g]->gtmf.p;
    bt.adym.p[g] = sds[g]->adym.p;
    bt.adymf.p[g] = sds[g]->adymf.p;
    bt.sval.p[g] = sds[g]->sval.p;
    bt.syval.p[g] = sds[g]->syval.p;
    bt.syvalb.p[g] = sds[g]->syvalb.p;
    bt.svalb.p[g] = sds[g]->svalb.p;
    bt.bvinv.p[g] = sds[g]->bvinv.p;
    bt.bpinv.p[g] = sds[g]->bpinv.p;
    bt.einv.p[g] = sds[g]->einv.p;
  }
  bt.gs = (size_t)c->n * m;
  bt.gsp = (size_t)c->np * m;
  bt.gsc = (size_t)c->kc * m;
  bt.gsq = (size_t)std::max(c->q, 1) * m;
  bt.all();
  if (c->child) {
    ShiftData* subs[RICADI_MAX_GROUPS];
    for (int g = 0; g < G; ++g) subs[g] = sds[g]->sub;
    bt.sub = std::make_shared<Batch>(make_batch(c->child.get(), subs, G, m));
  }
  return bt;
}
static Batch make_batch(ricadi_ctx* c, ShiftData* sd, int m) { return make_batch(c, &sd, 1, m); }

// Multi-shift tile kernel or one workgroup per (row block, group)?  The multi-shift kernel
// reads the matrix once for all groups (26 -> 18 B per non-zero in total instead of 10 B per
// group) but walks the groups of a row block one after the other at 4 waves per SIMD; it
// pays where the per-shift value arrays of the active groups no longer fit the caches
// (measured: n = 5e5, 16 groups: 1.53 -> 1.25 ms per launch; n = 3e4: 83 -> 87 us).
static bool ms_pays(const ricadi_ctx* c, int ng, size_t nnz) {
  if (!c->ms_spmm) return false;
  if (c->ms_force) return true;
  // per-shift value arrays of the active groups near or beyond the 256 MB infinity cache (measured with the FP32
  // operator input that follows this switch: cfg3, 227 MB: 197 -> 205 shift-solves/s; cfg2, 136 MB: 1.4 % slower)
  return ng >= 4 && (double)nnz * 10.0 * ng > 200e6;
}

// ---- operator and preconditioner on device panels ---------------------------------
// y = beta_r * r + alpha * S x on the saddle operator (optionally through the
// prolongation map): LDS-tiled kernel when the block tiles fit, else the CSR one.
// gsx / gsy / gsr: group strides of x, y, r.
// The LDS-tiled kernels serve panels of width m (else the CSR kernel runs)
static bool saddle_tiled(const ricadi_ctx* c, int m) {
  return c->sb_ok &&
         spmm_blocked_lds_bytes(m, c->sb_max_cols, c->sb_max_nnz) <= (size_t)40 * 1024;
}
// x32 (optional): FP32 copy of x with the same leading dimension and group stride; the tiled kernels read it
// instead of x (plain products only: no residual term, no low-rank epilogue, no prolongation map)
static void saddle_spmm(ricadi_ctx* c, const Batch& bt, const double* x, size_t gsx,
                        const int* xmap, double* y, size_t gsy, const double* r, size_t gsr,
                        double alpha, double beta_r, const LowRankArgs& lr = LowRankArgs(),
                        const float* x32 = nullptr, float* y32 = nullptr) {
  const int m = bt.m;
  const bool fits = saddle_tiled(c, m);
  const bool has_lr = lr.q > 0 && lr.nrows > 0;
  if (x32 && fits && !r && !xmap && !has_lr) {
    const bool ms = ms_pays(c, bt.tab.ng, c->snnz) && spmm_blocked_ms_ok(m, c->sb_max_cols, (size_t)c->n);
    c->k1_variant = (ms ? 2 : 1) + 4;
    if (ms)
      launch_spmm_blocked_ms_x32(c->st, bt.tab, bt.alpha, bt.beta, c->sb_nblk, c->sb_rows2.p, c->sb_rp2.p,
                                 c->sb_cols2.p, c->sb_lidx_ms.p, c->sbAJ.p, c->sbE.p, x32, m, gsx, y, m, gsy, alpha,
                                 m, c->sb_max_cols, y32);
    else
      launch_spmm_blocked_x32(c->st, bt.tab, c->sb_nblk, c->sb_rows2.p, c->sb_rp2.p, c->sb_cols2.p, c->sb_lidx.p,
                              bt.svalb, x32, m, gsx, y, m, gsy, alpha, m, c->sb_max_cols, y32);
    return;
  }
  if (y32) throw HipError{"FP32 operator output asked for outside the FP32-input tile kernels"};
  const bool ms = fits && ms_pays(c, bt.tab.ng, c->snnz) && !xmap && !has_lr &&
                  spmm_blocked_ms_ok(m, c->sb_max_cols, (size_t)c->n);
  if (!xmap) c->k1_variant = ms ? 2 : fits ? 1 : 0;
  if (ms)
    launch_spmm_blocked_ms(c->st, bt.tab, bt.alpha, bt.beta, c->sb_nblk, c->sb_rows2.p, c->sb_rp2.p,
                           c->sb_cols2.p, c->sb_lidx_ms.p, c->sbAJ.p, c->sbE.p, x, m, gsx, y, m, gsy,
                           r, m, gsr, alpha, beta_r, m, c->sb_max_cols);
  else if (fits)
    launch_spmm_blocked_b(c->st, bt.tab, c->sb_nblk, c->sb_rows2.p, c->sb_rp2.p,
                          xmap ? c->sb_colsm2.p : c->sb_cols2.p, c->sb_lidx.p, bt.svalb, x, m, gsx, y,
                          m, gsy, r, m, gsr, alpha, beta_r, m, c->sb_max_cols, lr);
  else
    launch_spmm_b(c->st, bt.tab, c->n, c->s_rp.p, c->s_ci.p, bt.sval, x, m, gsx, xmap, y, m, gsy, r,
                  m, gsr, alpha, beta_r, m, lr);
}

// Does the GMRES iteration apply the operator to the FP32-stored Z_j?
static bool operator_reads_x32(const ricadi_ctx* c, int m) {
  return c->flex && saddle_tiled(c, m);
}
// ... for a batch of ng groups: always with the multi-shift kernel; with one workgroup per (row block, group) the FP32
// input by itself measured 1.4 % slower at cfg2 in round 3, but it is what lets the cycle keep its velocity part in
// FP32 and its blocks in BF16 (round 4), which more than pays for it (RICADI_X32=0: only with the multi-shift kernel)
static bool iteration_reads_x32(const ricadi_ctx* c, int m, int ng) {
  return operator_reads_x32(c, m) && (c->x32_always || ms_pays(c, ng, c->snnz));
}

// y = S(alpha,beta) x for every active group (n x m panels, ld = m, group stride gsx /
// bt.gs); optional low-rank  - U V^T x_v  (U, V shared by the groups)
// y32 (optional, with x32 only): the product goes to this FP32 panel (stride bt.gs) and y is not written
static void op_apply(ricadi_ctx* c, const Batch& bt, const double* x, size_t gsx, double* y,
                     bool lowrank, const float* x32 = nullptr, float* y32 = nullptr) {
  hipStream_t st = c->st;
  const int m = bt.m;
  LowRankArgs lr;
  if (lowrank && c->q > 0) {
    // coefficients V^T x first; the product with U rides in the SpMM's epilogue
    HIPCHK(hipMemsetAsync(c->lrc.p, 0, sizeof(double) * bt.gsq * bt.G, st));
    launch_gemm_tn_b(st, bt.tab, c->nv, c->q, m, c->V.p, c->q, x, m, gsx, c->lrc.p, m, bt.gsq);
    lr.U = c->U.p;
    lr.c = c->lrc.p;
    lr.gsc = bt.gsq;
    lr.q = c->q;
    lr.nrows = c->nv;
  }
  saddle_spmm(c, bt, x, gsx, nullptr, y, bt.gs, nullptr, 0, 1.0, 0.0, lr, x32, y32);
}
// Does the Arnoldi iteration of a batch keep w = S z_j as an FP32 panel?  (the tile kernels with FP32 input write it,
// the three 16-column passes on the FP16-stored basis read it; RICADI_W32=0: FP64 panel)
static bool iteration_w32(const ricadi_ctx* c, int m, int ng, bool b16, bool fuseh, bool keepw, int restart) {
  return c->w32 && m == 16 && b16 && fuseh && keepw && iteration_reads_x32(c, m, ng) && arnoldi16_w32_ok(restart);
}

// z = P^-1 r for every active group: multiplicative two-level, coarse correction
// first, then one consistent SIMPLE block-Jacobi sweep on the updated residual.
// r has group stride gsr; z lives in a workspace buffer (stride bt.gs).
// z32 (optional, group stride gs32): FP32 copy of z, written by the sweeps that write z last.
// only32: z itself need not be stored where the sweeps write the copy (the operator will read z32).
// r16: the same residual panel as stored in FP16 (the current Krylov vector; group stride gsr); where the folded
// path runs, its three readers of r take the 2-byte copy (exactly the same values) and r itself is not touched.
static bool precond_folds(const ricadi_ctx* c) {
  return c->kc > 0 && c->ady_ok && c->np > 0;
}
static bool precond_reads_h16_static(const ricadi_ctx* c) {
  return precond_folds(c);
}
// Does the GMRES iteration hand the preconditioner the FP16-stored vector (RICADI_H16=0: the FP64 copy)?
static bool precond_reads_h16(const ricadi_ctx* c, int m) {
  return c->basis16 && m <= 16 && precond_folds(c);
}
static void precond_apply(ricadi_ctx* c, const Batch& bt, const double* r, size_t gsr, double* z,
                          float* z32 = nullptr, size_t gs32 = 0, bool only32 = false,
                          const _Float16* r16 = nullptr) {
  hipStream_t st = c->st;
  bool mirrored = false;
  const int nv = c->nv, np = c->np, m = bt.m;
  const GroupTab& gt = bt.tab;
  const GroupPtrs ones = same_ptr(c->ones.p), jv = same_ptr(c->J.v.p), jtv = same_ptr(c->JT.v.p);
  const double* rr = r;
  size_t gsrr = gsr;
  bool folded = false;
  // ricadi_time_kernel_dev times one stage at a time through exactly these launchers (c->pc_stage >= 0)
  auto on = [&](int stage) { return c->pc_stage < 0 || c->pc_stage == stage; };
  // the pressure step -- pressure rows of r - (S Y) e, J product, Schur sweep -- as ONE launch (K2p) for
  // 16-column panels (RICADI_PFUSE=0: the three launches of round 2)
  const bool fusedp = np > 0 && m == 16 && c->bs == 32;
  // The velocity part between the three sweeps (first sweep -> pressure step's J product -> last sweep) as an FP32
  // panel: where only the FP32 copy of z is wanted anyway (the operator reads Z_j as stored), the first sweep writes
  // the velocity rows of z32 itself, the pressure step gathers 64-B instead of 128-B rows and the last sweep updates
  // them in place.  Rounding the intermediate to FP32 perturbs the (flexible) preconditioner by what its FP32
  // inverses and the FP32-stored Z_j already do.
  const bool mid32 = c->mid32 && z32 && only32 && fusedp && precond_folds(c) && c->gt_ok && c->precond32;
  c->mid32_last = mid32 ? 1 : 0;
  // ... and on BF16-stored blocks where every shift of the batch has them (record-driven sweeps only)
  const bool b16 = mid32 && bt.blocks16 && c->sw_stride > 0 && c->bs == 32;
  if (c->kc > 0) {
    // restriction Y^T r = CSR product with unit values (aggregate lists as rows)
    folded = precond_folds(c);
    if (!folded || m > 16) r16 = nullptr;
    // (smoothed aggregation: P^T r with the rows of P^T)
    const int* rrp = c->sa ? c->pt_rp.p : c->agg_ptr.p;
    const int* rci = c->sa ? c->pt_ci.p : c->agg_rows.p;
    const GroupPtrs rvals = c->sa ? same_ptr((const double*)c->pt_v.p) : ones;
    if (c->sa && !folded) throw HipError{"smoothed aggregation needs the folded preconditioner cycle"};
    const size_t rnnz = c->sa ? c->pt_ci.n : (size_t)c->n;
    if (!on(0)) {
    } else if (m == 16 && c->rowwave && spmm_rowwave_pays(c->kc, rnnz))
      launch_spmm_rowwave(st, gt, c->kc, rrp, rci, rvals, r16 ? nullptr : r, r16, gsr, c->rc.p, bt.gsc, m);
    else if (r16)
      launch_spmm_h(st, gt, c->kc, rrp, rci, rvals, nullptr, r16, m, gsr, c->rc.p, m, bt.gsc,
                    nullptr, 0, 0, 1.0, 0.0, m, 16);
    else
      launch_spmm_b(st, gt, c->kc, rrp, rci, rvals, r, m, gsr, nullptr, c->rc.p, m,
                    bt.gsc, nullptr, 0, 0, 1.0, 0.0, m);
    if (!on(1)) {
    } else if (c->child) {
      // coarse problem by one cycle of the child level's preconditioner (a fixed linear operator)
      Batch cb = *bt.sub;
      cb.tab = gt;
      precond_apply(c->child.get(), cb, c->rc.p, bt.gsc, c->ec.p);
    } else if (c->precond32)
      launch_dense_apply_b(st, gt, c->kc, m, bt.einvf, (c->kc + 3) & ~3, c->rc.p, c->ec.p, c->coarse_mfma32);
    else
      launch_dense_apply_b(st, gt, c->kc, m, bt.einv, c->rc.p, c->ec.p);
    if (!on(2) || (fusedp && folded)) {
    } else if (folded) {
      // only the PRESSURE rows of r - (S Y) ec are formed (short CSR product over np rows); the
      // velocity rows ride inside the first velocity sweep (block_apply2_kernel, below)
      if (r16)
        launch_spmm_h(st, gt, np, c->sy_rp.p + nv, c->sy_ci.p, bt.syval, c->ec.p, nullptr, m, bt.gsc,
                      c->r2.p + (size_t)nv * m, m, bt.gs, r16 + (size_t)nv * m, m, gsr, -1.0, 1.0, m, c->sy_chunk);
      else
        launch_spmm_b(st, gt, np, c->sy_rp.p + nv, c->sy_ci.p, bt.syval, c->ec.p, m, bt.gsc, nullptr,
                      c->r2.p + (size_t)nv * m, m, bt.gs, r + (size_t)nv * m, m, gsr, -1.0, 1.0, m, LowRankArgs(),
                      c->sy_chunk);
    } else {
      // Residual after the coarse correction, r2 = r - (S Y) ec, with the prolongated
      // operator (short rows over the L2-resident coarse vector) -- not a full saddle SpMM
      // through the prolongation map.  (Forming the velocity rows of r2 inside the first
      // velocity sweep instead, like the J^T product below, was measured slower: 249 vs
      // 257 shift-solves/s -- 8 rows x 7.6 dependent gathers per lane.)
      // Tile form: the aggregates a row block touches (a few dozen coarse rows) go to LDS once.
      if (c->syb_ok && ms_pays(c, gt.ng, c->snnz) &&
          spmm_blocked_ms_ok(m, c->syb_max_cols, (size_t)c->kc))
        launch_spmm_blocked_ms(st, gt, bt.alpha, bt.beta, c->sb_nblk, c->sb_rows2.p, c->syb_rp2.p,
                               c->syb_cols2.p, c->syb_lidx_ms.p, c->sybAJ.p, c->sybE.p, c->ec.p, m,
                               bt.gsc, c->r2.p, m, bt.gs, r, m, gsr, -1.0, 1.0, m, c->syb_max_cols);
      else if (c->syb_ok &&
          spmm_blocked_lds_bytes(m, c->syb_max_cols, 0) <= (size_t)40 * 1024)
        launch_spmm_blocked_b(st, gt, c->sb_nblk, c->sb_rows2.p, c->syb_rp2.p, c->syb_cols2.p,
                              c->syb_lidx.p, bt.syvalb, c->ec.p, m, bt.gsc, c->r2.p, m, bt.gs, r, m, gsr,
                              -1.0, 1.0, m, c->syb_max_cols);
      else
        launch_spmm_b(st, gt, c->n, c->sy_rp.p, c->sy_ci.p, bt.syval, c->ec.p, m, bt.gsc, nullptr,
                      c->r2.p, m, bt.gs, r, m, gsr, -1.0, 1.0, m, LowRankArgs(), c->sy_chunk);
    }
    rr = c->r2.p;
    gsrr = bt.gs;
  }
  // the LAST velocity sweep also adds the coarse correction Y ec to all of z
  // (its surplus waves take the pressure rows)
  ProlongArgs pro;
  if (c->kc > 0) {
    pro.aggof = c->aggof.p;
    pro.ec = c->ec.p;
    pro.gse = bt.gsc;
    pro.row0 = nv;
    pro.nextra = np;
  }
  auto vel_apply = [&](const double* in, size_t gsi, int subtract, bool last,
                       const CsrInArgs& cin = CsrInArgs()) {
    const ProlongArgs pa = last ? pro : ProlongArgs();
    if (c->precond32)
      launch_block_apply_b(st, gt, c->bs, c->nbv, c->bv_ptr.p, c->bv_rows.p, bt.bvinvf, in, m, gsi, z,
                           m, bt.gs, m, subtract, pa, cin);
    else
      launch_block_apply_b(st, gt, c->bs, c->nbv, c->bv_ptr.p, c->bv_rows.p, bt.bvinv, in, m, gsi, z,
                           m, bt.gs, m, subtract, pa, cin);
  };
  if (!on(3)) {
  } else if (folded) {
    // z_v = Ahat^-1 r_v - (Ahat^-1 D) ec : first velocity sweep on the corrected residual without
    // ever writing it
    Seg2 s1, s2;
    s1.kstride = c->bs;
    s1.in = r16 ? nullptr : r;
    s1.in16 = r16;
    s1.gs = gsr;
    s2.iptr = c->cy_ptr.p;
    s2.irows = c->cy_cols.p;
    s2.kstride = c->ady_ks;
    s2.in = c->ec.p;
    s2.gs = bt.gsc;
    ProlongArgs fpa;
    if (mid32) {
      fpa.out32 = z32;
      fpa.gs32 = gs32;
      fpa.only32 = 1;
    }
    if (c->sw_stride > 0 && c->bs == 32) {
      fpa.bmeta = c->sw_meta.p;
      fpa.bm_stride = c->sw_stride;
      fpa.bm_in = c->sw_in_two;
      fpa.bm_ni = 2;
    }
    if (b16 && launch_block_two32_h(st, gt, c->nbv, bt.bvinvh, s1, bt.adymh, s2, z, bt.gs, fpa, c->sweep_mfma32)) {
    } else if (c->precond32)
      launch_block_apply2_b(st, gt, c->bs, c->nbv, c->bv_ptr.p, c->bv_rows.p, bt.bvinvf, s1, bt.adymf, s2, z, m,
                            bt.gs, m, fpa);
    else
      launch_block_apply2_b(st, gt, c->bs, c->nbv, c->bv_ptr.p, c->bv_rows.p, bt.bvinv, s1, bt.adym, s2, z, m,
                            bt.gs, m, fpa);
  } else {
    vel_apply(rr, gsrr, 0, np == 0);
  }
  if (np > 0) {
    // t = J z_v - r_p
    if (on(4) && !fusedp)
      launch_spmm_b(st, gt, np, c->J.rp.p, c->J.ci.p, jv, z, m, bt.gs, nullptr, c->tp.p, m, bt.gsp,
                    rr + (size_t)nv * m, m, gsrr, 1.0, -1.0, m);
    double* zp = z + (size_t)nv * m;
    // Fused variant: the pressure sweep writes z_p already WITH its coarse part and keeps
    // the plain z_p (the operand of the J^T product below) in tp -- in place: a wave
    // reads its block's rows of tp before it writes them, blocks are disjoint.
    ProlongArgs ppro;
    {
      ppro.out2 = c->tp.p;
      ppro.gs2 = bt.gsp;
      if (z32) {
        ppro.out32 = z32 + (size_t)nv * m;
        ppro.gs32 = gs32;
        ppro.only32 = only32 && c->gt_ok;   // the rectangle sweep below completes the FP32 copy
      }
      if (c->kc > 0) {
        ppro.aggof = c->aggof.p + nv;
        ppro.ec = c->ec.p;
        ppro.gse = bt.gsc;
      }
    }
    if (!on(5)) {
    } else if (fusedp) {
      // r_p: of the folded cycle the input vector itself (FP64 or FP16-stored) with the coarse term formed in
      // the kernel; else the pressure rows of the corrected residual r2
      const bool sy = folded;
      const double* rp64 = sy ? (r16 ? nullptr : r + (size_t)nv * m) : rr + (size_t)nv * m;
      const _Float16* rp16 = sy && r16 ? r16 + (size_t)nv * m : nullptr;
      const size_t gsrp = sy ? gsr : gsrr;
      if (b16)
        launch_pressure_step_h(st, gt, c->nbp, c->ps_meta.p, bt.bpinvh, c->J.ci.p, c->J.v.p, sy, c->sy_ci.p, bt.syval,
                               c->ec.p, bt.gsc, rp64, rp16, gsrp, zp, bt.gs, ppro, z32, gs32);
      else if (c->precond32)
        launch_pressure_step_b(st, gt, c->nbp, c->ps_meta.p, bt.bpinvf, c->J.ci.p, c->J.v.p, z, bt.gs, sy, c->sy_ci.p,
                               bt.syval, c->ec.p, bt.gsc, rp64, rp16, gsrp, zp, bt.gs, ppro, mid32 ? z32 : nullptr,
                               gs32);
      else
        launch_pressure_step_b(st, gt, c->nbp, c->ps_meta.p, bt.bpinv, c->J.ci.p, c->J.v.p, z, bt.gs, sy, c->sy_ci.p,
                               bt.syval, c->ec.p, bt.gsc, rp64, rp16, gsrp, zp, bt.gs, ppro);
    } else if (c->precond32)
      launch_block_apply_b(st, gt, c->bs, c->nbp, c->bp_ptr.p, c->bp_rows.p, bt.bpinvf, c->tp.p, m,
                           bt.gsp, zp, m, bt.gs, m, 0, ppro);
    else
      launch_block_apply_b(st, gt, c->bs, c->nbp, c->bp_ptr.p, c->bp_rows.p, bt.bpinv, c->tp.p, m,
                           bt.gsp, zp, m, bt.gs, m, 0, ppro);
    // z_v -= Ahat^-1 (J^T z_p): the same block-Jacobi inverse as in the Schur blocks; the
    // J^T product is formed inside the sweep, row by row as the blocks gather them
    // (z_p is small and L2 resident), instead of through an intermediate panel
    if (!on(6)) {
    } else if (c->gt_ok) {
      // z_v -= G z_p with the per-shift blocks G_b = Ahat_b^-1 J^T[rows_b, pcols_b] formed at setup
      pro.nextra = 0;            // the pressure rows already carry their coarse part
      pro.out32 = z32;
      pro.gs32 = gs32;
      pro.only32 = only32;
      pro.old32 = mid32 ? 1 : 0;
      if (c->sw_stride > 0 && c->bs == 32) {
        pro.bmeta = c->sw_meta.p;
        pro.bm_stride = c->sw_stride;
        pro.bm_in = c->sw_in_rect;
        pro.bm_ni = 1;
      }
      mirrored = true;
      if (b16 && launch_block_rect32_h(st, gt, c->gt_ks, c->nbv, bt.gtmh, c->tp.p, bt.gsp, z, bt.gs, 1, pro)) {
      } else if (c->precond32)
        launch_block_apply_rect_b(st, gt, c->bs, c->gt_ks, c->nbv, c->bv_ptr.p, c->bv_rows.p, c->gt_ptr.p,
                                  c->gt_cols.p, bt.gtmf, c->tp.p, m, bt.gsp, z, m, bt.gs, m, 1, pro);
      else
        launch_block_apply_rect_b(st, gt, c->bs, c->gt_ks, c->nbv, c->bv_ptr.p, c->bv_rows.p, c->gt_ptr.p,
                                  c->gt_cols.p, bt.gtm, c->tp.p, m, bt.gsp, z, m, bt.gs, m, 1, pro);
    } else {
      // blocks that touch too many pressure dofs for the dense rectangles: the J^T product formed row by row inside
      // the sweep (CsrInArgs)
      CsrInArgs cin;
      cin.rp = c->JT.rp.p;
      cin.ci = c->JT.ci.p;
      cin.v = jtv;
      cin.src = c->tp.p;
      cin.gss = bt.gsp;
      pro.nextra = 0;          // the pressure rows already carry their coarse part
      vel_apply(nullptr, 0, 1, true, cin);
    }
  }
  if (z32 && !mirrored && c->pc_stage < 0)
    for (int i = 0; i < gt.ng; ++i)
      launch_to_f32(st, c->n, m, z + (size_t)gt.gid[i] * bt.gs, m, z32 + (size_t)gt.gid[i] * gs32, m);
}

static void op_apply(ricadi_ctx* c, ShiftData* sd, const double* x, double* y, int m, bool lowrank) {
  const Batch bt = make_batch(c, sd, m);
  op_apply(c, bt, x, bt.gs, y, lowrank);
}
static void precond_apply(ricadi_ctx* c, ShiftData* sd, const double* r, double* z, int m) {
  const Batch bt = make_batch(c, sd, m);
  precond_apply(c, bt, r, bt.gs, z);
}

static void col_norms2(ricadi_ctx* c, const double* w, int nrows, int m, double* out) {
  launch_cols_dots(c->st, nrows, m, 0, nullptr, 0, w, 1, c->partial.p, out);
}

