// solver_gmres.inl -- lockstep batched GMRES, wide panels as column groups, recycled guesses, storage safety net, Sherman-Morrison-Woodbury.
// Part of ricadi_solver.hip (one translation unit; included there in order).

// ---- batched panel GMRES --------------------------------------------------------------
// Solves S(shift_g) x_g = b_g for the m columns of every group's n x m panel:
// one Arnoldi process per column, all groups in lockstep inside ONE sequence of
// launches (grid.z = active groups).  At n ~ 3e4 a single panel leaves most of
// the chip idle and the launch path dominates; batching the shifts of a sweep
// fills it.  Right preconditioning, CGS2, per-column Givens QR.  A group whose
// columns have all converged leaves the active table; its correction is formed
// at the end of the restart cycle from the basis vectors it had by then.
//   b: group stride gsb (0 = one right-hand side shared by all groups);
//   x: group stride n*m, overwritten.
struct GmresResult {
  int iters = 0;
  bool converged = false;
  bool stalled = false;       // gave up before gmres_maxit: three full-length cycles in a row gained < 30 %
  double max_relres = 0.0;
};

// have_x0: x holds an initial guess (else it is zeroed);  only: the groups to iterate on (NULL = all; the
// panels of the other groups are not touched);  allow_stall: a group whose full-length restart cycles no
// longer gain is given up early (the caller repeats it with wider storage).
static void gmres_core(ricadi_ctx* c, ShiftData* const* sds, int G, const double* b, size_t gsb, double* x,
                       int m, bool lowrank, GmresResult* res, bool have_x0, const std::vector<int>* only,
                       bool allow_stall) {
  ensure_work(c, m, G, 0);
  hipStream_t st = c->st;
  const int n = c->n, restart = c->opts.gmres_restart, maxit = c->opts.gmres_maxit;
  const double tol = c->opts.gmres_tol;
  Batch bt = make_batch(c, sds, G, m);
  const size_t nm = bt.gs;             // one panel
  const size_t vs = nm * G;            // one Krylov vector of all groups
  const size_t gsh = (size_t)(restart + 2) * m;
  const size_t gspart = (size_t)dots_num_blocks(n) * (restart + 2) * m;
  const int GM = G * m;
  double* V = c->basis.p;          // FP64 basis (RICADI_BASIS64) ...
  float* Vf = c->basisf.p;         // ... or the FP32-stored one
  const bool b16 = c->basis16;
  const bool b32 = c->basis32 && !b16;
  const bool flex = c->flex;
  // (only where the launches are bandwidth bound -- the multi-shift SpMM regime: cfg5 K1 1252 -> 1150 us per
  // launch, cycle +2 %; at cfg2 the FP32 gathers are no faster and the step was 1.4 % slower)
  // the preconditioner reads the current vector from the FP16 basis itself; its FP64 copy is then not written
  const bool h16 = precond_reads_h16(c, m);
  // dot passes with atomic accumulation (no partial rows, no reduce launches): FP16 basis, 16 columns
  // w is not rewritten between the two Gram-Schmidt passes: the final update subtracts V (h1 + h2) from the original w
  const bool keepw = update_dots_keeps_w(m, b16, restart);
  // last Arnoldi pass and Hessenberg update in ONE launch (K3h)
  const bool fuseh = update_hess_fused_ok(m, b16);
  const size_t resbuf = (size_t)c->wcols;                    // doubles between the two residual-estimate buffers
  struct NoStoreScope {
    explicit NoStoreScope(bool v) { set_update_dots_nostore(v); }
    ~NoStoreScope() { set_update_dots_nostore(false); }
  } nostore_scope(keepw);
  const size_t h2buf = (size_t)(restart + 2) * c->wcols;        // doubles between the two second-pass buffers
  const bool x32 = iteration_reads_x32(c, m, G) && !(lowrank && c->q > 0);
  const bool w32 = x32 && iteration_w32(c, m, G, b16, fuseh, keepw, restart);
  c->w32_last = w32 ? 1 : 0;
  _Float16* Vh = reinterpret_cast<_Float16*>(c->basisf.p);   // FP16 storage shares the FP32 buffer
  double* hb = c->h_resid;
  const size_t slot = (size_t)RICADI_MAX_M * RICADI_MAX_GROUPS;
  for (int g = 0; g < G; ++g) res[g] = GmresResult();

  auto norms2 = [&](const double* w, size_t gsw, double* out) {
    launch_cols_dots_b(st, bt.tab, n, m, 0, (const double*)nullptr, 0, 0, w, gsw, 1, c->partial.p, gspart, out,
                       (size_t)m);
  };
  bt.all();
  norms2(b, gsb, c->bnorm2.p);
  HIPCHK(hipMemcpyAsync(hb + slot, c->bnorm2.p, sizeof(double) * GM, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  std::vector<double> bn(GM);
  for (int j = 0; j < GM; ++j) bn[j] = std::sqrt(std::max(hb[slot + j], 0.0));
  // device copy of the norms (not squared) for the hess kernel
  HIPCHK(hipMemcpyAsync(c->bnorm2.p, bn.data(), sizeof(double) * GM, hipMemcpyHostToDevice, st));
  if (!have_x0) HIPCHK(hipMemsetAsync(x, 0, sizeof(double) * vs, st));

  auto group_converged = [&](const double* r, int g) {
    double worst = 0.0;
    bool ok = true;
    for (int j = g * m; j < (g + 1) * m; ++j) {
      const double rel = bn[j] > 0.0 ? r[j] / bn[j] : 0.0;
      worst = std::max(worst, rel);
      if (!(r[j] <= tol * bn[j])) ok = false;
    }
    res[g].max_relres = worst;
    return ok;
  };

  std::vector<char> done(G, 0);
  std::vector<int> act, live, kk(G, 0), nstall(G, 0);
  if (only) act = *only;
  else
    for (int g = 0; g < G; ++g) act.push_back(g);
  bool first = !have_x0;
  // Cycle length: short cycles keep the Krylov basis (the dominant HBM traffic of an
  // iteration: three passes over it) small; a cycle that gains less than a factor 10
  // on some column lengthens the following ones, up to gmres_restart.
  int cyc = std::min(restart, 10);
  std::vector<double> rstart(GM, 0.0);
  Tick tkc;
  auto lapc = [&](double& acc) {
    if (c->timing) {
      (void)hipStreamSynchronize(st);
      acc += tkc.lap();
    }
  };
  while (!act.empty()) {
    lapc(c->t_iter);
    bt.set(act);
    // residual of the current iterates
    if (first) {
      if (gsb == nm) {
        HIPCHK(hipMemcpyAsync(c->wv.p, b, sizeof(double) * vs, hipMemcpyDeviceToDevice, st));
      } else {
        for (int g = 0; g < G; ++g)
          HIPCHK(hipMemcpyAsync(c->wv.p + (size_t)g * nm, b + (size_t)g * gsb, sizeof(double) * nm,
                                hipMemcpyDeviceToDevice, st));
      }
    } else if (lowrank && c->q > 0) {
      op_apply(c, bt, x, nm, c->wv.p, lowrank);
      launch_axpby_b(st, bt.tab, nm, 1.0, b, gsb, -1.0, c->wv.p, nm);
    } else {
      // r = b - S x in one launch (the residual form of the SpMM)
      saddle_spmm(c, bt, x, nm, nullptr, c->wv.p, nm, b, gsb, -1.0, 1.0);
    }
    first = false;
    norms2(c->wv.p, nm, c->nrm2.p);
    launch_gmres_start_b(st, bt.tab, m, restart, c->nrm2.p, c->g.p, c->scale.p, c->resid.p);
    HIPCHK(hipMemcpyAsync(hb, c->resid.p, sizeof(double) * GM, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<int> next;
    bool slow = false;
    for (int g : act) {
      if (group_converged(hb, g)) {
        res[g].converged = true;
        done[g] = 1;
      } else if (res[g].iters >= maxit) {
        done[g] = 1;
      } else {
        bool flat = false;
        for (int j = g * m; j < (g + 1) * m; ++j) {
          if (rstart[j] > 0.0 && hb[j] > tol * bn[j] && hb[j] > 0.1 * rstart[j]) slow = true;
          if (rstart[j] > 0.0 && hb[j] > tol * bn[j] && hb[j] > 0.7 * rstart[j]) flat = true;
          rstart[j] = hb[j];
        }
        nstall[g] = (flat && cyc >= restart) ? nstall[g] + 1 : 0;
        if (allow_stall && nstall[g] >= 3) {
          res[g].stalled = true;
          done[g] = 1;
        } else {
          next.push_back(g);
        }
      }
    }
    if (slow) cyc = std::min(restart, cyc + (cyc + 1) / 2);
    // Few groups left (the stragglers of the sweep): the launches are latency bound then and
    // the traffic of a longer Krylov basis costs nothing -- let the cycles run to the full
    // restart length instead of throwing the subspace away every `cyc` vectors.
    act.swap(next);
    if (act.empty()) break;
    bt.set(act);
    if (b16)
      launch_colscale_b(st, bt.tab, n, m, c->scale.p, c->wv.p, nm, 0.0, c->vcur.p, nm, Vh, nm);
    else if (b32)
      launch_colscale_b(st, bt.tab, n, m, c->scale.p, c->wv.p, nm, 0.0, c->vcur.p, nm, Vf, nm);
    else
      launch_colscale_b(st, bt.tab, n, m, c->scale.p, c->wv.p, nm, 0.0, V, nm);
    live = act;
    for (int g : act) kk[g] = 0;
    lapc(c->t_cyc);
    for (int j = 0; j < cyc && !live.empty(); ++j) {
      bt.set(live);
      const double* vj = (b32 || b16) ? c->vcur.p : V + (size_t)j * vs;
      // flexible form: Z_j = P^-1 v_j is kept (FP32), the cycle's correction is x += Z y -- no
      // preconditioner application at the cycle end, and P may differ from step to step
      // ... and the operator reads that stored FP32 copy (half the bytes of the x gathers; S Z_j = V H then
      // holds for exactly the vectors the correction uses), so the sweeps need not store the FP64 z at all
      float* zj = flex ? c->zbasisf.p + (size_t)j * vs : nullptr;
      precond_apply(c, bt, vj, nm, c->zv.p, zj, nm, x32, h16 ? Vh + (size_t)j * vs : nullptr);
      op_apply(c, bt, c->zv.p, nm, c->wv.p, lowrank, x32 ? zj : nullptr, w32 ? c->wv32.p : nullptr);
      double* h2cur = c->h2.p;
      if (w32) {
        launch_cols_dots16_w32(st, bt.tab, n, j + 1, Vh, vs, nm, c->wv32.p, nm, c->partial.p, gspart, c->h1.p, gsh);
        launch_cols_update_dots16_w32(st, bt.tab, n, j + 1, Vh, vs, nm, c->h1.p, gsh, c->wv32.p, nm, c->partial.p,
                                      gspart, c->h2.p, gsh);
      } else if (b16) {
        launch_cols_dots_b(st, bt.tab, n, m, j + 1, Vh, vs, nm, c->wv.p, nm, 0, c->partial.p, gspart,
                           c->h1.p, gsh);
        launch_cols_update_dots_b(st, bt.tab, n, m, j + 1, Vh, vs, nm, c->h1.p, gsh, c->wv.p, nm,
                                  c->partial.p, gspart, c->h2.p, gsh);
      } else if (b32) {
        launch_cols_dots_b(st, bt.tab, n, m, j + 1, Vf, vs, nm, c->wv.p, nm, 0, c->partial.p, gspart,
                           c->h1.p, gsh);
        launch_cols_update_dots_b(st, bt.tab, n, m, j + 1, Vf, vs, nm, c->h1.p, gsh, c->wv.p, nm,
                                  c->partial.p, gspart, c->h2.p, gsh);
      } else {
        launch_cols_dots_b(st, bt.tab, n, m, j + 1, V, vs, nm, c->wv.p, nm, 0, c->partial.p, gspart,
                           c->h1.p, gsh);
        // first update fused with the dot products of the second pass
        launch_cols_update_dots_b(st, bt.tab, n, m, j + 1, V, vs, nm, c->h1.p, gsh, c->wv.p, nm,
                                  c->partial.p, gspart, c->h2.p, gsh);
      }
      // the residual estimates also go straight to a pinned host slot (read one
      // iteration later, behind the event below)
      double* cur = hb + 2 * slot + (size_t)(j & 1) * slot;
      if (fuseh)
        launch_cols_update16_hess_b(st, bt.tab, n, j + 1, Vh, vs, nm, c->h1.p, h2cur, gsh, keepw ? 1 : 0, c->wv.p, nm,
                                    h16 ? nullptr : c->vcur.p, nm, Vh + (size_t)(j + 1) * vs, nm, j, restart, c->H.p,
                                    c->cs.p, c->sn.p, c->g.p, c->resid.p + (size_t)(j & 1) * resbuf,
                                    c->resid.p + (size_t)((j + 1) & 1) * resbuf, c->bnorm2.p, tol, cur,
                                    w32 ? c->wv32.p : nullptr);
      else
        launch_gmres_hess_b(st, bt.tab, m, j, restart, c->h1.p, h2cur, c->H.p, c->cs.p, c->sn.p,
                            c->g.p, c->scale.p, c->resid.p, c->bnorm2.p, tol, cur, nullptr, nullptr,
                            keepw ? c->h2.p + h2buf : nullptr);
      if (fuseh) {
      } else if (b16)
        launch_cols_update_b(st, bt.tab, n, m, j + 1, Vh, vs, nm, keepw ? c->h2.p + h2buf : h2cur, gsh, -1.0, c->wv.p, nm,
                             c->scale.p, h16 ? nullptr : c->vcur.p, nm, Vh + (size_t)(j + 1) * vs, nm);
      else if (b32)
        launch_cols_update_b(st, bt.tab, n, m, j + 1, Vf, vs, nm, c->h2.p, gsh, -1.0, c->wv.p, nm,
                             c->scale.p, c->vcur.p, nm, Vf + (size_t)(j + 1) * vs, nm);
      else
        launch_cols_update_b(st, bt.tab, n, m, j + 1, V, vs, nm, c->h2.p, gsh, -1.0, c->wv.p, nm,
                             c->scale.p, V + (size_t)(j + 1) * vs, nm);
      // Residual estimates travel to a pinned slot behind an event; the host
      // looks at the PREVIOUS iteration's slot, so it never drains the stream
      // (one iteration of lag: at most one surplus Arnoldi step per group).
      HIPCHK(hipEventRecord(c->ev_res[j & 1], st));
      for (int g : live) {
        ++res[g].iters;
        kk[g] = j + 1;
      }
      std::vector<int> still;
      if (j >= 1) {
        HIPCHK(hipEventSynchronize(c->ev_res[(j - 1) & 1]));
        const double* prev = hb + 2 * slot + (size_t)((j - 1) & 1) * slot;
        for (int g : live)
          if (!group_converged(prev, g) && res[g].iters < maxit) still.push_back(g);
      } else {
        for (int g : live)
          if (res[g].iters < maxit) still.push_back(g);
      }
      live.swap(still);
    }
    lapc(c->t_iter);
    // corrections: x_g += P^-1 (V_g y_g) with the k_g basis vectors group g built
    // (one launch each for all groups of the cycle, k_g per group by value)
    bt.set(act);
    {
      GroupInts ks = same_int(0);
      for (int g : act) ks.v[g] = kk[g];
      launch_gmres_backsolve_b(st, bt.tab, m, ks, restart, c->H.p, c->g.p, c->yv.p);
      if (flex)      // x += Z y in one launch
        launch_cols_update_bk(st, bt.tab, n, m, ks, c->zbasisf.p, vs, nm, c->yv.p, (size_t)restart * m, x, nm, x, nm);
      else if (b16)
        launch_cols_update_bk(st, bt.tab, n, m, ks, Vh, vs, nm, c->yv.p, (size_t)restart * m, c->wv.p, nm);
      else if (b32)
        launch_cols_update_bk(st, bt.tab, n, m, ks, Vf, vs, nm, c->yv.p, (size_t)restart * m, c->wv.p, nm);
      else
        launch_cols_update_bk(st, bt.tab, n, m, ks, V, vs, nm, c->yv.p, (size_t)restart * m, c->wv.p, nm);
    }
    bt.set(act);
    if (flex) {
    } else {
      precond_apply(c, bt, c->wv.p, nm, c->zv.p);
      launch_axpby_b(st, bt.tab, nm, 1.0, c->zv.p, nm, 1.0, x, nm);
    }
    lapc(c->t_cyc);
  }
  lapc(c->t_cyc);
}

// ---- wide panels as sixteen-column groups -------------------------------------------------------
// The columns of a panel are independent Arnoldi processes (per-column Givens), so an n x m panel with
// m > 32 -- the time-varying Riccati loop's [M^T Z_c, sqrt(tau) C~^T, K_k] of up to comprz_maxc + NY' + NU
// columns, /root/reference/solve_dae_ric.py:149 -- is solved as groups of 16 columns of the SAME shift in
// the lockstep batch: every kernel tuned for the 16-column case (LDS-tiled SpMM, 16-byte Arnoldi kernels,
// fused pressure step, FP16 vector input) then carries the iteration instead of the generic-width ones.
// The shifts of the call are walked in chunks of floor(RICADI_MAX_GROUPS / groups per shift); the column
// groups are scattered into / gathered from group-major panels (pad columns are zero: a zero column is
// inert in every kernel of the iteration).  RICADI_WIDE_SPLIT=0 keeps the wide panels whole.
static void gmres_core_any(ricadi_ctx* c, ShiftData* const* sds, int G, const double* b, size_t gsb, double* x,
                           int m, bool lowrank, GmresResult* res, bool have_x0, const std::vector<int>* only,
                           bool allow_stall) {
  const int W0 = wide_split_width(c, m);
  if (!W0) {
    gmres_core(c, sds, G, b, gsb, x, m, lowrank, res, have_x0, only, allow_stall);
    return;
  }
  hipStream_t st = c->st;
  const int n = c->n;
  const size_t nm = (size_t)n * m;
  std::vector<int> todo;
  if (only) todo = *only;
  else
    for (int g = 0; g < G; ++g) todo.push_back(g);
  for (int g = 0; g < G; ++g) res[g] = GmresResult();
  for (int s : todo) res[s].converged = true;
  // (ensure_work of the caller reserved a full chunk; growing the workspace here would free the buffer b lives in)
  if (c->wcols < W0 * RICADI_MAX_GROUPS) throw HipError{"workspace not sized for the column groups of a wide panel"};
  c->split_b.ensure((size_t)n * W0 * RICADI_MAX_GROUPS);
  c->split_x.ensure((size_t)n * W0 * RICADI_MAX_GROUPS);
  // columns [col0, col0 + ncols) of every panel as groups of W columns
  auto run_pass = [&](int col0, int ncols, int W) {
    const int ncg = (ncols + W - 1) / W;
    const int per = std::max(1, RICADI_MAX_GROUPS / ncg);
    const size_t nmw = (size_t)n * W;
    // chunks of equal size (16 shifts, 3 per chunk: 3 3 3 3 2 2 rather than 3 3 3 3 3 1); the caller's order is
    // kept: neighbouring shifts of a sorted list need similar iteration counts, which is what a lockstep batch wants
    const int nchunk = ((int)todo.size() + per - 1) / per;
    size_t at = 0;
    for (int ch = 0; ch < nchunk; ++ch) {
      const int cnt = ((int)todo.size() - (int)at + (nchunk - ch) - 1) / (nchunk - ch);
      const int Gv = cnt * ncg;
      std::vector<ShiftData*> vsds(Gv);
      if (ncg * W != ncols) {
        HIPCHK(hipMemsetAsync(c->split_b.p, 0, sizeof(double) * nmw * Gv, st));
        if (have_x0) HIPCHK(hipMemsetAsync(c->split_x.p, 0, sizeof(double) * nmw * Gv, st));
      }
      for (int k = 0; k < cnt; ++k) {
        const int s = todo[at + k];
        for (int cg = 0; cg < ncg; ++cg) {
          const int v = k * ncg + cg, w = std::min(W, ncols - cg * W), sc = col0 + cg * W;
          vsds[v] = sds[s];
          launch_copy_cols(st, n, w, b + (size_t)s * gsb, m, sc, c->split_b.p + (size_t)v * nmw, W, 0, 1.0);
          if (have_x0)
            launch_copy_cols(st, n, w, x + (size_t)s * nm, m, sc, c->split_x.p + (size_t)v * nmw, W, 0, 1.0);
        }
      }
      std::vector<GmresResult> vres(Gv);
      gmres_core(c, vsds.data(), Gv, c->split_b.p, nmw, c->split_x.p, W, lowrank, vres.data(), have_x0, nullptr,
                 allow_stall);
      for (int k = 0; k < cnt; ++k) {
        const int s = todo[at + k];
        GmresResult& r = res[s];
        for (int cg = 0; cg < ncg; ++cg) {
          const int v = k * ncg + cg, w = std::min(W, ncols - cg * W);
          launch_copy_cols(st, n, w, c->split_x.p + (size_t)v * nmw, W, 0, x + (size_t)s * nm, m, col0 + cg * W, 1.0);
          r.iters = std::max(r.iters, vres[v].iters);
          r.converged = r.converged && vres[v].converged;
          r.stalled = r.stalled || vres[v].stalled;
          r.max_relres = std::max(r.max_relres, vres[v].max_relres);
        }
      }
      at += cnt;
    }
  };
  // (a remainder of up to 8 columns -- m = 66 = 4 x 16 + 2 -- as one more batch of 8-column groups over all
  // shifts instead of a fifth sixteen-column group per shift was measured at n = 1e5: 2172 vs 2176 ms per pass
  // over 64 shifts; the sweeps of an 8-column batch cost what those of a 16-column one do -- the block inverses
  // they read are as many bytes as the panels)
  run_pass(0, m, W0);
}

// ---- recycled right-hand sides (ricadi_set_recycle) ---------------------------------------------
// Initial guesses  x_g = sum_e Y_{g,e} C_e  from the stored pairs (B_e, Y_{g,e}),  S_g Y_{g,e} = B_e, with
// C = argmin || b - [B_e] C ||_F  (normal equations on the matrix cores, rank-revealing Cholesky on the
// host).  b: the right-hand side shared by the groups (n x m, pressure rows zero).  Returns false when no
// stored panel is common to all groups (x is not touched then).
static bool recycle_guess(ricadi_ctx* c, ShiftData* const* sds, int G, const double* b, int m, double* x) {
  std::vector<const ricadi_ctx::RecB*> ent;
  for (auto& e : c->rec_ring) {
    if (!e || e->serial < 0) continue;
    bool all = true;
    for (int g = 0; g < G && all; ++g) {
      bool has = false;
      for (auto& y : sds[g]->rec)
        if (y && y->serial == e->serial && y->w == e->w) has = true;
      all = has;
    }
    if (all) ent.push_back(e.get());
  }
  if (ent.empty()) return false;
  int h = 0;
  for (auto* e : ent) h += e->w;
  hipStream_t st = c->st;
  const int nv = c->nv, n = c->n, hw = h + m;
  TArr<double> Gd(c->pool), Yd(c->pool, (size_t)h * m);
  std::vector<double> Ghh((size_t)h * h), Ghb((size_t)h * m), Y;
  int r0 = 0;
  // slot of every entry in the side-by-side panel (all of the panel's width, ring of at most 8 slots)
  std::vector<int> slot_of(ent.size(), -1);
  bool pan = c->rec_pan_w == m && c->rec_pan.p && c->rec_ring.size() <= 8;
  for (size_t i = 0; i < ent.size() && pan; ++i) {
    for (size_t si = 0; si < c->rec_ring.size(); ++si)
      if (c->rec_ring[si].get() == ent[i]) slot_of[i] = (int)si;
    pan = slot_of[i] >= 0 && ent[i]->w == m;
  }
  if (pan) {
    // Gram matrix of ALL slots and their products with b in two launches; the live entries are picked on the host
    const int H = 8 * m, Hw = H + m;
    Gd.alloc((size_t)H * Hw);
    HIPCHK(hipMemsetAsync(Gd.p, 0, sizeof(double) * H * Hw, st));
    launch_gemm_tn(st, nv, H, H, c->rec_pan.p, H, c->rec_pan.p, H, Gd.p, Hw);
    launch_gemm_tn(st, nv, H, m, c->rec_pan.p, H, b, m, Gd.p + H, Hw);
    std::vector<double> Gh((size_t)H * Hw);
    HIPCHK(hipMemcpyAsync(Gh.data(), Gd.p, sizeof(double) * Gh.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (size_t ei = 0; ei < ent.size(); ++ei)
      for (int a = 0; a < m; ++a) {
        const int i = (int)ei * m + a, gi = slot_of[ei] * m + a;
        for (size_t ej = 0; ej < ent.size(); ++ej)
          for (int bcol = 0; bcol < m; ++bcol)
            Ghh[(size_t)i * h + ej * m + bcol] = Gh[(size_t)gi * Hw + slot_of[ej] * m + bcol];
        for (int j = 0; j < m; ++j) Ghb[(size_t)i * m + j] = Gh[(size_t)gi * Hw + H + j];
      }
  } else {
    Gd.alloc((size_t)h * hw);
    HIPCHK(hipMemsetAsync(Gd.p, 0, sizeof(double) * h * hw, st));
    for (size_t i = 0; i < ent.size(); ++i) {
      int c0 = r0;
      for (size_t j = i; j < ent.size(); ++j) {
        launch_gemm_tn(st, nv, ent[i]->w, ent[j]->w, ent[i]->b.p, ent[i]->w, ent[j]->b.p, ent[j]->w,
                       Gd.p + (size_t)r0 * hw + c0, hw);
        c0 += ent[j]->w;
      }
      launch_gemm_tn(st, nv, ent[i]->w, m, ent[i]->b.p, ent[i]->w, b, m, Gd.p + (size_t)r0 * hw + h, hw);
      r0 += ent[i]->w;
    }
    std::vector<double> Gh((size_t)h * hw);
    HIPCHK(hipMemcpyAsync(Gh.data(), Gd.p, sizeof(double) * Gh.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < h; ++i) {
      for (int j = 0; j < h; ++j) Ghh[(size_t)i * h + j] = j >= i ? Gh[(size_t)i * hw + j] : Gh[(size_t)j * hw + i];
      for (int j = 0; j < m; ++j) Ghb[(size_t)i * m + j] = Gh[(size_t)i * hw + h + j];
    }
  }
  // the diagonal blocks come from a symmetric kernel, the off-diagonal ones were computed above the
  // diagonal only: the mirror image is exact
  const int rank = gram_lstsq_scaled(h, m, Ghh, Ghb, 1e-11, Y);
  if (rank == 0) return false;
  HIPCHK(hipMemcpyAsync(Yd.p, Y.data(), sizeof(double) * h * m, hipMemcpyHostToDevice, st));
  GroupTab all{};
  all.ng = G;
  for (int g = 0; g < G; ++g) all.gid[g] = g;
  r0 = 0;
  for (size_t i = 0; i < ent.size(); ++i) {
    GroupPtrs A = same_ptr((const double*)nullptr);
    for (int g = 0; g < G; ++g)
      for (auto& y : sds[g]->rec)
        if (y && y->serial == ent[i]->serial && y->w == ent[i]->w) A.p[g] = y->y.p;
    launch_gemm_nn_bp(st, all, n, ent[i]->w, m, A, ent[i]->w, Yd.p + (size_t)r0 * m, m, 0, x, m, (size_t)n * m,
                      1.0, i == 0 ? 0.0 : 1.0);
    r0 += ent[i]->w;
  }
  HIPCHK(hipStreamSynchronize(st));   // Y is a stack object
  if (c->opts.verbose > 1) fprintf(stderr, "[ricadi] recycled guess from %d stored columns (rank %d)\n", h, rank);
  return true;
}

static void recycle_store(ricadi_ctx* c, ShiftData* const* sds, int G, const double* b, int m, const double* x) {
  hipStream_t st = c->st;
  const int depth = c->rec_depth;
  ricadi_ctx::RecB* slot = nullptr;
  if ((int)c->rec_ring.size() < depth) {
    c->rec_ring.emplace_back(new ricadi_ctx::RecB);
    slot = c->rec_ring.back().get();
  } else {
    for (auto& e : c->rec_ring)
      if (!slot || e->serial < slot->serial) slot = e.get();
  }
  slot->serial = ++c->rec_serial;
  slot->w = m;
  slot->b.ensure((size_t)c->nv * m);
  HIPCHK(hipMemcpyAsync(slot->b.p, b, sizeof(double) * c->nv * m, hipMemcpyDeviceToDevice, st));
  {
    // side-by-side copy (slots of another width invalidate the panel: recycle_guess then takes the pairwise path)
    int si = 0;
    for (; si < (int)c->rec_ring.size(); ++si)
      if (c->rec_ring[si].get() == slot) break;
    if (c->rec_pan_w != m || c->rec_pan.n < (size_t)c->nv * 8 * m) {
      c->rec_pan.ensure((size_t)c->nv * 8 * m);
      HIPCHK(hipMemsetAsync(c->rec_pan.p, 0, sizeof(double) * (size_t)c->nv * 8 * m, st));
      c->rec_pan_w = m;
      for (auto& e : c->rec_ring)
        if (e.get() != slot && e->serial >= 0 && e->w == m)
          launch_copy_cols(st, c->nv, m, e->b.p, m, 0, c->rec_pan.p, 8 * m, (int)(&e - &c->rec_ring[0]) * m, 1.0);
    }
    if (si < 8) launch_copy_cols(st, c->nv, m, b, m, 0, c->rec_pan.p, 8 * m, si * m, 1.0);
  }
  auto live = [&](long serial) {
    for (auto& e : c->rec_ring)
      if (e->serial == serial) return true;
    return false;
  };
  const size_t nm = (size_t)c->n * m;
  for (int g = 0; g < G; ++g) {
    ShiftData::RecY* y = nullptr;
    for (auto& r : sds[g]->rec)
      if (!live(r->serial)) y = r.get();          // a solution whose right-hand side has left the ring
    if (!y && (int)sds[g]->rec.size() < depth) {
      sds[g]->rec.emplace_back(new ShiftData::RecY);
      y = sds[g]->rec.back().get();
    }
    if (!y)
      for (auto& r : sds[g]->rec)
        if (!y || r->serial < y->serial) y = r.get();
    y->serial = slot->serial;
    y->w = m;
    y->y.ensure(nm);
    HIPCHK(hipMemcpyAsync(y->y.p, x + (size_t)g * nm, sizeof(double) * nm, hipMemcpyDeviceToDevice, st));
  }
}

// Storage of the Krylov basis / the preconditioner inverses for the solves inside the scope:
//   level 1: FP32-stored basis, FP64 inverses;  level 2: FP64-stored basis, FP64 inverses
// (level 0 = the context's defaults: FP16 / FP32 basis by size, FP32 inverses).  All levels of a
// multilevel preconditioner follow.  The arithmetic is FP64 at every level.
struct StorageScope {
  ricadi_ctx* c;
  bool b16, b32;
  std::vector<bool> p32;
  StorageScope(ricadi_ctx* ctx, int level) : c(ctx), b16(ctx->basis16), b32(ctx->basis32) {
    for (ricadi_ctx* l = c; l; l = l->child.get()) {
      p32.push_back(l->precond32);
      l->precond32 = false;
    }
    c->basis16 = false;
    if (level >= 2) {
      c->basis32 = false;
      c->basis.ensure((size_t)(c->wrestart + 1) * c->n * c->wcols);
    }
  }
  ~StorageScope() {
    size_t i = 0;
    for (ricadi_ctx* l = c; l; l = l->child.get()) l->precond32 = p32[i++];
    c->basis16 = b16;
    c->basis32 = b32;
  }
};
static int storage_level(const ricadi_ctx* c) {
  if (!c->precond32 && !c->basis32) return 2;
  if (!c->precond32 && !c->basis16) return 1;
  return 0;
}

// The batched solve as the drivers call it: recycled initial guess (shared right-hand side, plain
// operator), the lockstep GMRES, the storage safety net -- a group that stops at gmres_maxit or
// stagnates is continued from its iterate with the FP32- and then the FP64-stored basis and FP64
// preconditioner inverses (counted in c->escalations) -- true residuals on request.
static void gmres_solve_batch(ricadi_ctx* c, ShiftData* const* sds, int G, const double* b,
                              size_t gsb, double* x, int m, bool lowrank, double* relres_host,
                              GmresResult* res) {
  hipStream_t st = c->st;
  const bool plain = !(lowrank && c->q > 0);
  const bool shared = (gsb == 0 || G == 1) && plain && c->rec_depth > 0;
  Tick tkg;
  const bool guess = shared && recycle_guess(c, sds, G, b, m, x);
  if (c->timing) {
    (void)hipStreamSynchronize(st);
    c->t_guess += tkg.lap();
  }
  const int lvl0 = storage_level(c);
  gmres_core_any(c, sds, G, b, gsb, x, m, lowrank, res, guess, nullptr, lvl0 < 2);
  std::vector<int> bad;
  for (int g = 0; g < G; ++g)
    if (!res[g].converged) bad.push_back(g);
  for (int level = lvl0 + 1; level <= 2 && !bad.empty(); ++level) {
    StorageScope wide(c, level);
    std::vector<GmresResult> r2(G);
    gmres_core_any(c, sds, G, b, gsb, x, m, lowrank, r2.data(), true, &bad, level < 2);
    c->escalations += (long)bad.size();
    std::vector<int> still;
    for (int g : bad) {
      if (c->opts.verbose)
        fprintf(stderr, "[ricadi] shift (%g, %g): %s after %d iterations at relres %.2e -> storage level %d: %d more, %.2e\n",
                sds[g]->alpha, sds[g]->beta, res[g].stalled ? "stagnation" : "gmres_maxit", res[g].iters,
                res[g].max_relres, level, r2[g].iters, r2[g].max_relres);
      res[g].iters += r2[g].iters;
      res[g].converged = r2[g].converged;
      res[g].stalled = r2[g].stalled;
      res[g].max_relres = r2[g].max_relres;
      if (!r2[g].converged) still.push_back(g);
    }
    bad.swap(still);
  }
  if (relres_host) {
    // true residuals
    Batch bt = make_batch(c, sds, G, m);
    const size_t nm = bt.gs;
    const size_t gspart = (size_t)dots_num_blocks(c->n) * (c->opts.gmres_restart + 2) * m;
    const int GM = G * m;
    double* hb = c->h_resid;
    op_apply(c, bt, x, nm, c->wv.p, lowrank);
    launch_axpby_b(st, bt.tab, nm, 1.0, b, gsb, -1.0, c->wv.p, nm);
    launch_cols_dots_b(st, bt.tab, c->n, m, 0, (const double*)nullptr, 0, 0, c->wv.p, nm, 1, c->partial.p,
                       gspart, c->nrm2.p, (size_t)m);
    launch_cols_dots_b(st, bt.tab, c->n, m, 0, (const double*)nullptr, 0, 0, b, gsb, 1, c->partial.p,
                       gspart, c->bnorm2.p, (size_t)m);
    HIPCHK(hipMemcpyAsync(hb, c->nrm2.p, sizeof(double) * GM, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hb + GM, c->bnorm2.p, sizeof(double) * GM, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int j = 0; j < GM; ++j)
      relres_host[j] = hb[GM + j] > 0.0 ? std::sqrt(std::max(hb[j], 0.0) / hb[GM + j]) : 0.0;
  }
  if (shared) recycle_store(c, sds, G, b, m, x);
  for (int g = 0; g < G; ++g) c->total_iters += res[g].iters;
  c->total_solves += G;
}

// In-place inverse of a small dense matrix on the host (Gauss-Jordan, partial pivoting).
static bool host_invert(std::vector<double>& a, int q) {
  std::vector<double> inv((size_t)q * q, 0.0);
  for (int i = 0; i < q; ++i) inv[(size_t)i * q + i] = 1.0;
  double amax = 0.0;
  for (double v : a) amax = std::max(amax, std::fabs(v));
  for (int k = 0; k < q; ++k) {
    int p = k;
    for (int i = k + 1; i < q; ++i)
      if (std::fabs(a[(size_t)i * q + k]) > std::fabs(a[(size_t)p * q + k])) p = i;
    const double piv = a[(size_t)p * q + k];
    if (!(std::fabs(piv) > 1e-12 * amax)) return false;
    if (p != k)
      for (int j = 0; j < q; ++j) {
        std::swap(a[(size_t)k * q + j], a[(size_t)p * q + j]);
        std::swap(inv[(size_t)k * q + j], inv[(size_t)p * q + j]);
      }
    for (int j = 0; j < q; ++j) {
      a[(size_t)k * q + j] /= piv;
      inv[(size_t)k * q + j] /= piv;
    }
    for (int i = 0; i < q; ++i) {
      if (i == k) continue;
      const double f = a[(size_t)i * q + k];
      if (f == 0.0) continue;
      for (int j = 0; j < q; ++j) {
        a[(size_t)i * q + j] -= f * a[(size_t)k * q + j];
        inv[(size_t)i * q + j] -= f * inv[(size_t)k * q + j];
      }
    }
  }
  a.swap(inv);
  return true;
}

// Relative true residuals ||b - (S - U V^T) x|| / ||b|| per column (G*m values, host);
// the residual panels are left in c->wv.
static void true_relres(ricadi_ctx* c, ShiftData* const* sds, int G, const double* b, size_t gsb,
                        const double* x, int m, bool lowrank, double* out) {
  hipStream_t st = c->st;
  Batch bt = make_batch(c, sds, G, m);
  const size_t nm = bt.gs;
  const size_t gspart = (size_t)dots_num_blocks(c->n) * (c->opts.gmres_restart + 2) * m;
  const int GM = G * m;
  double* hb = c->h_resid;
  op_apply(c, bt, x, nm, c->wv.p, lowrank);
  launch_axpby_b(st, bt.tab, nm, 1.0, b, gsb, -1.0, c->wv.p, nm);
  launch_cols_dots_b(st, bt.tab, c->n, m, 0, (const double*)nullptr, 0, 0, c->wv.p, nm, 1, c->partial.p,
                     gspart, c->nrm2.p, (size_t)m);
  launch_cols_dots_b(st, bt.tab, c->n, m, 0, (const double*)nullptr, 0, 0, b, gsb, 1, c->partial.p,
                     gspart, c->bnorm2.p, (size_t)m);
  HIPCHK(hipMemcpyAsync(hb, c->nrm2.p, sizeof(double) * GM, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(hb + GM, c->bnorm2.p, sizeof(double) * GM, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for (int j = 0; j < GM; ++j)
    out[j] = hb[GM + j] > 0.0 ? std::sqrt(std::max(hb[j], 0.0) / hb[GM + j]) : 0.0;
}

// Batched solve with the low-rank term  (S_g - U V^T) x_g = b_g.
//
// Default: Sherman-Morrison-Woodbury, as the reference's lau.solve_sadpnt_smw does --
// GMRES runs on the plain saddle operator (no thin GEMMs inside the iteration), and
//   x = y + W (V^T y),   y = S^-1 b,   W = S^-1 [U;0] (I - V^T S^-1 U)^-1 .
// W_g is cached per shift and low-rank term; a batch that meets a shift without it
// solves the augmented panels [b_g, U] (m + q columns) once.  The closed-loop residual
// is then verified in FP64; columns above the tolerance (ill-conditioned capacitance
// matrix) are refined by one GMRES on the closed-loop operator itself.
static void solve_batch(ricadi_ctx* c, ShiftData* const* sds, int G, const double* b, size_t gsb,
                        double* x, int m, bool lowrank, double* relres_host, GmresResult* res) {
  const int q = c->q;
  if (!lowrank || q <= 0 || !c->smw || m + q > RICADI_MAX_M) {
    gmres_solve_batch(c, sds, G, b, gsb, x, m, lowrank && q > 0, relres_host, res);
    return;
  }
  hipStream_t st = c->st;
  const int n = c->n, nv = c->nv, np = c->np;
  const size_t nm = (size_t)n * m;
  const double tol = c->opts.gmres_tol;
  bool need = false;
  for (int g = 0; g < G; ++g) need = need || sds[g]->smw_epoch != c->lr_epoch;
  GroupTab all{};
  all.ng = G;
  for (int g = 0; g < G; ++g) all.gid[g] = g;
  bool bad = false;
  // U = columns [ucol, ucol + q) of the (shared) right-hand side: S^-1 U is part of the plain solution,
  // no augmented columns needed (first sweep of a Newton step without mtxoldb: rhs = [W, K_k], U = K_k)
  const int ucol = c->lr_ucol;
  c->lr_ucol = -1;             // the hint holds for one solve
  const bool dup = need && (gsb == 0 || G == 1) && ucol >= 0 && ucol + q <= m;
  if (need) {
    const int ma = dup ? m : m + q;
    const size_t nma = (size_t)n * ma;
    double* xa;
    int xoff;     // column of S^-1 U inside the solution panels xa (leading dimension ma)
    if (dup) {
      gmres_solve_batch(c, sds, G, b, gsb, x, m, false, nullptr, res);
      xa = x;
      xoff = ucol;
    } else {
      // augmented panels [b_g, U]; one panel for all groups when they share b
      const int nra = gsb == 0 ? 1 : G;
      c->smw_rhs.ensure(nma * nra);
      c->smw_x.ensure(nma * G);
      double* ra = c->smw_rhs.p;
      xa = c->smw_x.p;
      xoff = m;
      for (int g = 0; g < nra; ++g) {
        launch_copy_cols(st, n, m, b + (size_t)g * gsb, m, 0, ra + g * nma, ma, 0, 1.0);
        launch_copy_cols(st, nv, q, c->U.p, q, 0, ra + g * nma, ma, m, 1.0);
        if (np > 0)
          HIPCHK(hipMemset2DAsync(ra + g * nma + (size_t)nv * ma + m, sizeof(double) * ma, 0,
                                  sizeof(double) * q, np, st));
      }
      gmres_solve_batch(c, sds, G, ra, gsb == 0 ? 0 : nma, xa, ma, false, nullptr, res);
    }
    // capacitance matrices I - V^T (S^-1 U)
    c->smw_cap.ensure((size_t)G * q * q);
    HIPCHK(hipMemsetAsync(c->smw_cap.p, 0, sizeof(double) * G * q * q, st));
    launch_gemm_tn_b(st, all, nv, q, q, c->V.p, q, xa + xoff, ma, nma, c->smw_cap.p, q, (size_t)q * q);
    std::vector<double> caps((size_t)G * q * q);
    HIPCHK(hipMemcpyAsync(caps.data(), c->smw_cap.p, sizeof(double) * caps.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int g = 0; g < G && !bad; ++g) {
      std::vector<double> cap((size_t)q * q);
      for (int i = 0; i < q; ++i)
        for (int j = 0; j < q; ++j)
          cap[(size_t)i * q + j] = (i == j ? 1.0 : 0.0) - caps[(size_t)g * q * q + (size_t)i * q + j];
      if (!host_invert(cap, q) || !res[g].converged) bad = true;
      std::copy(cap.begin(), cap.end(), caps.begin() + (size_t)g * q * q);
    }
    if (!dup)
      for (int g = 0; g < G; ++g)
        launch_copy_cols(st, n, m, xa + g * nma, ma, 0, x + g * nm, m, 0, 1.0);
    if (!bad) {
      HIPCHK(hipMemcpyAsync(c->smw_cap.p, caps.data(), sizeof(double) * caps.size(), hipMemcpyHostToDevice, st));
      for (int g = 0; g < G; ++g) {
        ShiftData* sd = sds[g];
        if (sd->smw_w.n != (size_t)n * q) sd->smw_w.alloc((size_t)n * q);
        launch_gemm_nn(st, n, q, q, xa + g * nma + xoff, ma, c->smw_cap.p + (size_t)g * q * q, q,
                       sd->smw_w.p, q, 1.0, 0.0);
        sd->smw_epoch = c->lr_epoch;
      }
      HIPCHK(hipStreamSynchronize(st));   // caps is a stack object
    }
  } else {
    gmres_solve_batch(c, sds, G, b, gsb, x, m, false, nullptr, res);
  }
  const size_t gsq = (size_t)q * m;
  if (!bad) {
    // x_g += W_g (V^T x_g)
    GroupPtrs W = same_ptr((const double*)nullptr);
    for (int g = 0; g < G; ++g) W.p[g] = sds[g]->smw_w.p;
    HIPCHK(hipMemsetAsync(c->lrc.p, 0, sizeof(double) * gsq * G, st));
    launch_gemm_tn_b(st, all, nv, q, m, c->V.p, q, x, m, nm, c->lrc.p, m, gsq);
    launch_gemm_nn_bp(st, all, n, q, m, W, q, c->lrc.p, m, gsq, x, m, nm, 1.0, 1.0);
  }
  // verification on the closed-loop operator, refinement where needed
  std::vector<double> rr((size_t)G * m);
  true_relres(c, sds, G, b, gsb, x, m, true, rr.data());
  bool ok = true;
  for (double v : rr) ok = ok && v <= tol;
  if (!ok) {
    c->smw_rhs.ensure(nm * G);
    c->smw_x.ensure(nm * G);
    HIPCHK(hipMemcpyAsync(c->smw_rhs.p, c->wv.p, sizeof(double) * nm * G, hipMemcpyDeviceToDevice, st));
    std::vector<GmresResult> r2(G);
    // residual equation on the closed-loop operator; its tolerance is relative to ||r||
    double worst = 0.0;
    for (double v : rr) worst = std::max(worst, v);
    {
      Restore<double> keep_tol(c->opts.gmres_tol);
      c->opts.gmres_tol = std::min(0.5, std::max(1e-14, 0.5 * tol / worst));
      gmres_solve_batch(c, sds, G, c->smw_rhs.p, nm, c->smw_x.p, m, true, nullptr, r2.data());
    }
    launch_axpby_b(st, all, nm, 1.0, c->smw_x.p, nm, 1.0, x, nm);
    for (int g = 0; g < G; ++g) res[g].iters += r2[g].iters;
    true_relres(c, sds, G, b, gsb, x, m, true, rr.data());
  }
  for (int g = 0; g < G; ++g) {
    double w = 0.0;
    for (int j = 0; j < m; ++j) w = std::max(w, rr[(size_t)g * m + j]);
    res[g].max_relres = w;
    res[g].converged = w <= tol * 1.0000001;
  }
  if (relres_host) std::copy(rr.begin(), rr.end(), relres_host);
}

static GmresResult gmres_solve(ricadi_ctx* c, ShiftData* sd, const double* b, double* x, int m,
                               bool lowrank, double* relres_host) {
  GmresResult r;
  solve_batch(c, &sd, 1, b, (size_t)c->n * m, x, m, lowrank, relres_host, &r);
  return r;
}

// rhs panel (n x m) from an NV x m device block (pressure rows zero)
static void load_rhs(ricadi_ctx* c, const double* dR, int m, double* b) {
  HIPCHK(hipMemcpyAsync(b, dR, sizeof(double) * (size_t)c->nv * m, hipMemcpyDeviceToDevice, c->st));
  if (c->np > 0)
    HIPCHK(hipMemsetAsync(b + (size_t)c->nv * m, 0, sizeof(double) * (size_t)c->np * m, c->st));
}

// Per-shift data of the ADI shifts an iteration is about to use -- and of the projection
// operator (alpha, beta) = (1, 0) when `with_projection` -- built in ONE setup pass: the
// coarse matrices of all of them go through the same batched factorisation (a matrix set
// up alone costs ~8x its share of a batch of 16).
static void prefetch_setup(ricadi_ctx* c, const double* shifts, int nuse, bool with_projection) {
  std::vector<double> al, be;
  if (with_projection && c->np > 0) {
    al.push_back(1.0);
    be.push_back(0.0);
  }
  for (int i = 0; i < nuse; ++i) {
    al.push_back(shifts[i]);
    be.push_back(1.0);
  }
  if (al.empty()) return;
  std::vector<ShiftData*> sds(al.size());
  get_shifts(c, al.data(), be.data(), (int)al.size(), sds.data());
}

// W (NV x m, device, in place) <- P^T W  through one saddle solve with cal E
static void project_panel(ricadi_ctx* c, double* dW, int m) {
  if (c->np == 0) return;
  ShiftData* sd = get_shift(c, 1.0, 0.0);
  ensure_work(c, m);
  load_rhs(c, dW, m, c->bvec.p);
  GmresResult r = gmres_solve(c, sd, c->bvec.p, c->xs.p, m, false, nullptr);
  if (!r.converged) throw HipError{"projection solve did not converge"};
  launch_spmm(c->st, c->nv, c->E.rp.p, c->E.ci.p, c->E.v.p, c->xs.p, m, nullptr, dW, m, nullptr, 0,
              1.0, 0.0, nullptr, m);
}

struct DScalar {
  // tiny helper: Frobenius norm of W^T W and ||W||_F^2 of a device panel
  static void gram_norms(ricadi_ctx* c, const double* dW, int nrows, int m, double* gram_fro,
                         double* nrm2) {
    DArr<double>& G = c->scratch;
    G.ensure((size_t)m * m + 64);
    HIPCHK(hipMemsetAsync(G.p, 0, sizeof(double) * m * m, c->st));
    launch_gemm_tn(c->st, nrows, m, m, dW, m, dW, m, G.p, m);
    std::vector<double> h((size_t)m * m);
    HIPCHK(hipMemcpyAsync(h.data(), G.p, sizeof(double) * m * m, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    double f = 0.0, t = 0.0;
    for (int i = 0; i < m; ++i) {
      t += h[(size_t)i * m + i];
      for (int j = 0; j < m; ++j) f += h[(size_t)i * m + j] * h[(size_t)i * m + j];
    }
    if (gram_fro) *gram_fro = std::sqrt(f);
    if (nrm2) *nrm2 = t;
  }
};

static int compress_dev(ricadi_ctx* c, const double* dZ, int cz, int ldz, double thresh, int kmax,
                        bool thresh_relative, double* dOut, std::vector<double>* sv_host,
                        bool use_qr = false);
static void block_qr_dev(ricadi_ctx* c, const double* D, int ldd, int n, int kk, double* Q,
                         double* R, int split = 0);

// Truncation level of the internal recompressions: the Gram-matrix route
// resolves singular values down to sqrt(eps)*sigma_1; dropping what lies below
// changes Z Z^T by at most eps*||Z Z^T|| -- rounding level.
static const double kInternalRelThresh = 3e-8;

static Exec main_exec(ricadi_ctx* c);
static int recompress_exec(ricadi_ctx* c, const Exec& ex, const double* dZ, int cz, int ldz, double rel,
                           double* dOut);

// Recompress the device factor in place (columns [0, zc) of c->Z).
static void factor_recompress(ricadi_ctx* c) {
  if (c->zc == 0) return;
  TArr<double> tmp(c->pool, (size_t)c->nv * c->zc);
  const int k = recompress_exec(c, main_exec(c), c->Z.p, c->zc, c->zld, kInternalRelThresh, tmp.p);
  if (k > 0) launch_copy_cols(c->st, c->nv, k, tmp.p, k, 0, c->Z.p, c->zld, 0, 1.0);
  HIPCHK(hipStreamSynchronize(c->st));
  c->zc = k;
}

static int compress_gram_exec(ricadi_ctx* c, const Exec& ex, const double* dZ, int cz, int ldz,
                              double thresh, int kmax, bool thresh_relative, double* dOut,
                              std::vector<double>* sv_host);
static int recompress_exec(ricadi_ctx* c, const Exec& ex, const double* dZ, int cz, int ldz, double rel,
                           double* dOut);

// Auxiliary stream + handle for work that runs beside the main stream (created on first use).
static Exec aux_exec(ricadi_ctx* c) {
  if (!c->st2) {
    HIPCHK(hipStreamCreateWithFlags(&c->st2, hipStreamNonBlocking));
    RBCHK(rocblas_create_handle(&c->rb2));
    RBCHK(rocblas_set_stream(c->rb2, c->st2));
    c->info2.alloc(4);
    HIPCHK(hipEventCreateWithFlags(&c->ev_z, hipEventDisableTiming));
  }
  Exec ex;
  ex.st = c->st2;
  ex.rb = c->rb2;
  ex.pool = &c->pool2;
  ex.info = c->info2.p;
  return ex;
}

// In-ADI recompression that does not stall the sweeps: the columns [0, snap) of the factor are
// compressed on the auxiliary stream by a helper thread (same arithmetic as
// factor_recompress) while the main stream goes on appending columns behind them; finish()
// splices the result in:  Z <- [compressed prefix | columns appended meanwhile].
// Member order matters: `fut` is destroyed first and waits for the helper, then `out`.
struct AsyncRecompress {
  ricadi_ctx* c;
  TArr<double> out;
  int snap = 0;
  bool active = false;
  std::future<int> fut;
  explicit AsyncRecompress(ricadi_ctx* ctx) : c(ctx), out(ctx->pool) {}
  void start() {
    if (active || c->zc == 0) return;
    const Exec ex = aux_exec(c);
    snap = c->zc;
    out.alloc((size_t)c->nv * snap);
    HIPCHK(hipEventRecord(c->ev_z, c->st));            // the prefix is complete on the main stream
    HIPCHK(hipStreamWaitEvent(c->st2, c->ev_z, 0));
    ricadi_ctx* cc = c;
    const double* Zp = c->Z.p;
    const int ld = c->zld, sn = snap, dev = c->dev;
    double* op = out.p;
    fut = std::async(std::launch::async, [cc, ex, Zp, ld, sn, dev, op]() {
      (void)hipSetDevice(dev);
      return recompress_exec(cc, ex, Zp, sn, ld, kInternalRelThresh, op);
    });
    active = true;
  }
  void finish() {
    if (!active) return;
    active = false;
    const int k = fut.get();                            // the auxiliary stream is drained in there
    hipStream_t st = c->st;
    const int nv = c->nv, tail = c->zc - snap;
    if (tail > 0) {
      TArr<double> tmp(c->pool, (size_t)nv * tail);
      launch_copy_cols(st, nv, tail, c->Z.p, c->zld, snap, tmp.p, tail, 0, 1.0);
      launch_copy_cols(st, nv, tail, tmp.p, tail, 0, c->Z.p, c->zld, k, 1.0);
    }
    if (k > 0) launch_copy_cols(st, nv, k, out.p, k, 0, c->Z.p, c->zld, 0, 1.0);
    c->zc = k + tail;
    out.release();
  }
};

