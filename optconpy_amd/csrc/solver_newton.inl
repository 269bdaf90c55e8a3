// solver_newton.inl -- Newton-Kleinman iteration on device operands (pru.proj_alg_ric_newtonadi).
// Part of ricadi_solver.hip (one translation unit; included there in order).


// Newton-Kleinman iteration on DEVICE operands (all panels row-major, their own width as leading dimension):
// dB nv x nb, dW nv x mw, dZ0 nv x c0 (or c0 = 0), dOld nv x nb or NULL (`oldB`: whether it is given).  The new
// iterate is left in the context's factor (c->Z, c->zc).
static void ric_newtonadi_run(ricadi_ctx* c, const double* shifts, int ns, const double* dB, int nb, const double* dW,
                              int mw, const double* dZ0, int c0, const double* dOld, const ricadi_adi_params* prm,
                              double* stats_out) {
  const bool oldB = dOld != nullptr;
  hipStream_t st = c->st;
  const int nv = c->nv;
  const int mfull = mw + nb;
  // The Newton loop installs its own low-rank term (K_k - old) B^T in the context; whatever
  // way this function is left -- also by an exception -- no stale term may stay behind for
  // later ricadi_lyap_adi / ricadi_shift_solve calls.
  struct LowRankReset {
    ricadi_ctx* c;
    ~LowRankReset() {
      c->q = 0;
      ++c->lr_epoch;
    }
  } lowrank_reset{c};
  ensure_work(c, mfull);
  TArr<double> dWm(c->pool, (size_t)nv * mw), dK(c->pool, (size_t)nv * nb), dKall(c->pool, (size_t)nv * nb),
      dRhs(c->pool, (size_t)nv * mfull), Zown(c->pool), Znew(c->pool);
  // W is projected in place below: private copy; B, the old gain and Z0 are only read
  HIPCHK(hipMemcpyAsync(dWm.p, dW, sizeof(double) * nv * mw, hipMemcpyDeviceToDevice, st));
  const double* zk = c0 > 0 ? dZ0 : nullptr;       // current (compressed) iterate Z_k, nv x kk (ld kk)
  int kk = c0;
  // the rhs factor W is projected once here; the K_k part is in range(P^T) already
  ricadi_adi_params p2 = *prm;
  Tick tk0;
  prefetch_setup(c, shifts, std::min(ns, prm->adi_max_steps), prm->project_w != 0);
  const double t_pre = c->timing ? ((void)hipStreamSynchronize(st), tk0.lap()) : 0.0;
  if (prm->project_w) project_panel(c, dWm.p, mw);
  if (c->timing) {
    (void)hipStreamSynchronize(st);
    fprintf(stderr, "[ricadi timing] per-shift setup of %d shifts + projection operator %.1f ms, projection solve %.1f ms\n",
            std::min(ns, prm->adi_max_steps), 1e3 * t_pre, 1e3 * tk0.lap());
  }
  p2.project_w = 0;
  if (p2.compress_cols <= 0) {
    // columns the factor may grow by before it is recompressed (RICADI_COMPRESS_COLS overrides): rocSOLVER's
    // tridiagonalisation is launch bound at these sizes (~32 us per column), so fewer, larger eigenproblems are cheaper
    p2.compress_cols = 512;
  }
  double upd = 0, updrel = 0;
  long adi_total = 0, gm_total = 0, sol_total = 0, nonconv = 0, sweep_total = 0;
  const long esc0 = c->escalations;
  double worst = 0.0, last_res = 0.0, last_rhs = 0.0;
  int steps = 0;
  for (steps = 1; steps <= prm->nwtn_max_steps; ++steps) {
    int m = mw;
    Tick tkn;
    if (c->timing) c->t_setup = c->t_solve = c->t_recomb = c->t_compress = c->t_updnorm = c->t_proj = c->t_gain = c->t_cyc = c->t_iter = c->t_guess = 0;
    if (kk > 0) {
      gain_dev(c, c->E, zk, kk, kk, dB, nb, dK.p);
      if (c->timing) c->t_gain += tkn.lap();
      m = mfull;
    } else {
      HIPCHK(hipMemsetAsync(dK.p, 0, sizeof(double) * nv * nb, st));
    }
    // closed loop  cal A - (K_k - old) B^T
    HIPCHK(hipMemcpyAsync(dKall.p, dK.p, sizeof(double) * nv * nb, hipMemcpyDeviceToDevice, st));
    if (oldB) launch_axpby(st, (size_t)nv * nb, -1.0, dOld, 1.0, dKall.p);
    const bool lr = (kk > 0) || oldB;
    c->q = lr ? nb : 0;
    ++c->lr_epoch;
    if (lr) {
      c->U.ensure((size_t)nv * nb);
      c->V.ensure((size_t)nv * nb);
      HIPCHK(hipMemcpyAsync(c->U.p, dKall.p, sizeof(double) * nv * nb, hipMemcpyDeviceToDevice, st));
      HIPCHK(hipMemcpyAsync(c->V.p, dB, sizeof(double) * nv * nb, hipMemcpyDeviceToDevice, st));
    }
    // rhs = [W, K_k]
    launch_copy_cols(st, nv, mw, dWm.p, mw, 0, dRhs.p, m, 0, 1.0);
    if (m > mw) launch_copy_cols(st, nv, nb, dK.p, nb, 0, dRhs.p, m, mw, 1.0);
    factor_reserve(c, prm->adi_max_steps * m);
    DScalar::gram_norms(c, dRhs.p, nv, m, &last_rhs, nullptr);
    // without mtxoldb the low-rank factor U = K_k is the last nb columns of the rhs itself
    c->lr_ucol = (lr && !oldB && m > mw) ? mw : -1;
    AdiStats s = lyap_adi_dev(c, shifts, ns, dRhs.p, m, p2);
    c->lr_ucol = -1;
    last_res = s.res_fro;
    adi_total += s.steps;
    gm_total += s.gmres_iters;
    sol_total += s.shift_solves;
    sweep_total += s.sweeps;
    nonconv += s.nonconverged;
    worst = std::max(worst, s.worst_relres);
    // compressed copy of the new iterate (truncation at the Gram noise floor)
    const int zraw = c->zc;
    Tick tkc;
    factor_recompress(c);
    if (c->timing) c->t_compress += tkc.lap();
    Znew.alloc((size_t)nv * c->zc);
    const int knew = c->zc;
    launch_copy_cols(st, nv, knew, c->Z.p, c->zld, 0, Znew.p, knew, 0, 1.0);
    double x1 = 0.0;
    upd = diff_zzt_fnorm(c, Znew.p, knew, zk, kk, &x1);
    updrel = x1 > 0.0 ? upd / x1 : 0.0;
    {
      double dec[2] = {upd, updrel};       // the stopping decision is rank 0's
      values_of_rank0(c, dec, 2);
      upd = dec[0];
      updrel = dec[1];
    }
    if (c->timing) {
      c->t_updnorm += tkc.lap();
      fprintf(stderr, "[ricadi timing] inside the solves: Arnoldi iterations %.1f ms, restart-cycle bookkeeping %.1f, recycled guesses %.1f\n",
              1e3 * c->t_iter, 1e3 * c->t_cyc, 1e3 * c->t_guess);
      fprintf(stderr, "[ricadi timing] Newton step %d: total %.1f ms = setup %.1f + projection %.1f + solves %.1f + "
              "recombination %.1f + recompression %.1f + update norm %.1f + gain %.1f (+ rest); %d raw columns at the end\n",
              steps, 1e3 * tkn.lap(), 1e3 * c->t_setup, 1e3 * c->t_proj, 1e3 * c->t_solve, 1e3 * c->t_recomb,
              1e3 * c->t_compress, 1e3 * c->t_updnorm, 1e3 * c->t_gain, zraw);
    }
    if (prm->verbose)
      fprintf(stderr, "[ricadi] Newton step %2d: |upd| %9.3e rel %9.3e (%d ADI steps, %d -> %d columns)\n",
              steps, upd, updrel, s.steps, c->zc, knew);
    Zown.swap(Znew);
    zk = Zown.p;
    kk = knew;
    if (upd < prm->nwtn_upd_abstol || updrel < prm->nwtn_upd_reltol) break;
  }
  if (steps > prm->nwtn_max_steps) steps = prm->nwtn_max_steps;
  if (stats_out) {
    stats_out[0] = steps;
    stats_out[1] = upd;
    stats_out[2] = updrel;
    stats_out[3] = (double)adi_total;
    stats_out[4] = (double)gm_total;
    stats_out[5] = (double)sol_total;
    stats_out[6] = (double)nonconv;
    stats_out[7] = worst;
    stats_out[8] = last_res;
    stats_out[9] = last_rhs;
    stats_out[10] = (double)(c->escalations - esc0);
    stats_out[11] = (double)sweep_total;
  }
}

