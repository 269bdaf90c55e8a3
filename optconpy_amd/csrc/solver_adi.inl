// solver_adi.inl -- low-rank ADI in step and sweep form, rank-sharded sweeps (exchange), Newton-Kleinman driver.
// Part of ricadi_solver.hip (one translation unit; included there in order).

// ---- low-rank ADI (device resident) -------------------------------------------------
struct AdiStats {
  int steps = 0;
  double rel = 0.0;
  long gmres_iters = 0;
  long shift_solves = 0;
  double res_fro = 0.0;
  long nonconverged = 0;      // shift-solves that hit gmres_maxit above the tolerance
  double worst_relres = 0.0;
  int sweeps = 0;             // sweep form: batched sweeps run (= all-gathers when sharded)
};

// dW: NV x m device panel (overwritten by the final residual factor).
// Appends sqrt(-2p) V_i to c->Z (ld = c->zld) starting at column c->zc.
// Sweep form of the same ADI: G consecutive steps
// with distinct shifts are G independent solves against the SAME residual factor,
//   S(p_g) [U_g; *] = [W; 0],
// recombined with the G x G Cauchy matrix C_ij = -1/(p_i+p_j) = R^T R:
//   Z-block = U (R^-1 (x) I),   W <- W + E U ((C^-1 1) (x) I)
// -- identical to the G sequential steps up to a rotation of the block's columns (Z Z^T
// and the gain are the same).  The G solves go through ONE batched lockstep GMRES, which
// is what fills the GPU at n ~ 3e4.  The stopping rule is applied per sweep (mean block
// norm).  Returns false (nothing done) if the shift list does not allow sweeps.
// All-gather of `count` doubles per rank through the host's collective (ricadi_set_exchange): the ranks'
// first `count` doubles of c->xsend arrive rank-major in c->xrecv.  The context stream is drained first.
// The last RICADI_XCTL bytes of the send buffer (and the last world * RICADI_XCTL of the receive buffer) are
// kept for the small control messages (decisions, statistics), so that they never touch panels in flight.
#define RICADI_XCTL 4096
static size_t exchange_panel_capacity(const ricadi_ctx* c) { return c->xcap > RICADI_XCTL ? c->xcap - RICADI_XCTL : 0; }
static void exchange_at(ricadi_ctx* c, double* send, double* recv, size_t count) {
  ++c->xcount;
  if (c->xcomm) {
    // RCCL: stream ordered behind the solves that filled `send`, ahead of the recombination that reads `recv`
    const ncclResult_t r = ncclAllGather(send, recv, count, ncclDouble, c->xcomm, c->st);
    if (r != ncclSuccess) throw HipError{std::string("ncclAllGather: ") + ncclGetErrorString(r)};
    return;
  }
  HIPCHK(hipStreamSynchronize(c->st));
  const int rc = c->xfn(c->xuser, send, recv, (int64_t)(count * sizeof(double)));
  if (rc != 0) throw HipError{"the all-gather callback of ricadi_set_exchange failed (" + std::to_string(rc) + ")"};
}
static void exchange(ricadi_ctx* c, size_t count) {
  if (count * sizeof(double) > exchange_panel_capacity(c))
    throw HipError{"exchange buffer too small: " + std::to_string(count * sizeof(double) + RICADI_XCTL) +
                   " bytes per rank needed, " + std::to_string(c->xcap) + " given to ricadi_set_exchange"};
  exchange_at(c, c->xsend, c->xrecv, count);
}
static double* ctl_send(ricadi_ctx* c) { return c->xsend + exchange_panel_capacity(c) / sizeof(double); }
static double* ctl_recv(ricadi_ctx* c) {
  return c->xrecv + (size_t)c->xworld * exchange_panel_capacity(c) / sizeof(double);
}
static bool sharded(const ricadi_ctx* c) { return (c->xworld > 1 || c->xforce) && (c->xfn != nullptr || c->xcomm != nullptr); }
// v[0..n) <- rank 0's values (decisions must not differ between the ranks: the norms they rest on come
// from kernels with atomic accumulation).  One tiny all-gather.
static void values_of_rank0(ricadi_ctx* c, double* v, int n) {
  if (!sharded(c)) return;
  if ((size_t)n * sizeof(double) > RICADI_XCTL) throw HipError{"control message too long"};
  HIPCHK(hipMemcpyAsync(ctl_send(c), v, sizeof(double) * n, hipMemcpyHostToDevice, c->st));
  exchange_at(c, ctl_send(c), ctl_recv(c), (size_t)n);
  HIPCHK(hipMemcpyAsync(v, ctl_recv(c), sizeof(double) * n, hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
}
// v[0..nsum) <- sum over the ranks, v[nsum..nsum+nmax) <- maximum over the ranks (statistics)
static void reduce_over_ranks(ricadi_ctx* c, double* v, int nsum, int nmax) {
  if (!sharded(c)) return;
  const int n = nsum + nmax;
  if ((size_t)n * sizeof(double) > RICADI_XCTL) throw HipError{"control message too long"};
  HIPCHK(hipMemcpyAsync(ctl_send(c), v, sizeof(double) * n, hipMemcpyHostToDevice, c->st));
  exchange_at(c, ctl_send(c), ctl_recv(c), (size_t)n);
  std::vector<double> all((size_t)n * c->xworld);
  HIPCHK(hipMemcpyAsync(all.data(), ctl_recv(c), sizeof(double) * all.size(), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  for (int i = 0; i < n; ++i) {
    double t = all[i];
    for (int r = 1; r < c->xworld; ++r) {
      const double o = all[(size_t)r * n + i];
      t = i < nsum ? t + o : std::max(t, o);
    }
    v[i] = t;
  }
}

static bool lyap_adi_sweeps_dev(ricadi_ctx* c, const double* shifts, int ns, double* dW, int m,
                                const ricadi_adi_params& prm, AdiStats& stt) {
  int G = std::min(std::min(prm.sweep_width, ns), RICADI_MAX_GROUPS);
  while (G >= 2 && (prm.adi_max_steps < G || G * m > 2048)) G /= 2;
  if (G < 2) return false;
  for (int i = 0; i < ns; ++i)
    for (int j = i + 1; j < ns; ++j)
      if (shifts[i] == shifts[j]) return false;     // sweeps need distinct shifts
  // Cauchy data of every sweep of the cycle up front.  A numerically singular Cauchy matrix (the shifts of a
  // sweep too many / too close: 16 consecutive entries of a 128-shift list over 3.5 decades) halves the sweep
  // width until every sweep of the cycle is admissible -- as shift_parallel.py does --; only when no width
  // >= 2 is left does the caller go back to the sequential form
  int ncyc = 0;
  std::vector<std::vector<double>> rinvs, cinvs, pss;
  for (; G >= 2; G /= 2) {
    ncyc = ns / std::gcd(ns, G);                  // sweeps until the shift pattern repeats
    rinvs.assign(ncyc, {});
    cinvs.assign(ncyc, {});
    pss.assign(ncyc, {});
    bool ok = true;
    for (int sw = 0; sw < ncyc && ok; ++sw) {
      pss[sw].resize(G);
      for (int g = 0; g < G; ++g) pss[sw][g] = shifts[(sw * G + g) % ns];
      rinvs[sw].resize((size_t)G * G);
      cinvs[sw].resize(G);
      ok = cauchy_data(pss[sw].data(), G, rinvs[sw].data(), cinvs[sw].data()) == RICADI_OK;
    }
    if (ok) break;
  }
  if (G < 2) return false;
  hipStream_t st = c->st;
  const int n = c->n, nv = c->nv;
  const size_t nm = (size_t)n * m;
  // Shift-parallel form (ricadi_set_exchange): every rank owns a fixed subset of the shift list --
  // fixed, because the per-shift setup, the Sherman-Morrison-Woodbury panels and the recycled
  // solutions live with the owner -- and solves only its shifts of a sweep; one all-gather per sweep.
  const bool shard = sharded(c);
  const int world = shard ? c->xworld : 1, rank = shard ? c->xrank : 0;
  std::vector<int32_t> owner(ns, 0);
  if (shard && deal_shifts(shifts, ns, world, owner.data()) != RICADI_OK) throw HipError{"bad shift list"};
  ensure_work(c, m, G);
  Tick tk;
  auto lap = [&](double& acc) {
    if (c->timing) {
      (void)hipStreamSynchronize(st);
      acc += tk.lap();
    }
  };
  // A failure in the OWNER-LOCAL work of a rank (per-shift setup: a singular block; its solves) must not leave the
  // other ranks waiting in the sweep's all-gather: it is recorded here, the rank still takes part in the exchange
  // -- with zero panels and its status word set --, and all ranks throw together once the words have gone round.
  // The words ride in the pressure rows of each rank's first solution panel (the recombination reads velocity
  // rows only), so a sweep costs ONE collective.
  std::string fail;
  auto guarded = [&](auto&& body) {
    if (!shard) {
      body();
      return;
    }
    try {
      body();
    } catch (const HipError& e) {
      fail = e.msg;
    } catch (const std::exception& e) {
      fail = e.what();
    }
  };
  const bool words_fit = (size_t)c->np * m >= 2;
  guarded([&] {
    std::vector<double> mine;
    const int nuse = std::min(ns, prm.adi_max_steps);
    for (int i = 0; i < nuse; ++i)
      if (owner[i] == rank) mine.push_back(shifts[i]);
    prefetch_setup(c, mine.data(), (int)mine.size(), prm.project_w != 0);
  });
  lap(c->t_setup);
  if (prm.project_w) {
    // (replicated: the projection operator is set up by every rank; a rank whose own setup failed skips it)
    if (fail.empty()) guarded([&] { project_panel(c, dW, m); });
  }
  lap(c->t_proj);
  const long it0 = c->total_iters;
  if (!shard) c->sweep_u.ensure(nm * G);
  c->sweep_t.ensure((size_t)nv * m);
  double znorm2 = 0.0;
  int zc_last = c->zc;
  std::vector<double> be(G, 1.0), coef;
  std::vector<ShiftData*> sds(G);
  std::vector<GmresResult> res(G);
  AsyncRecompress job(c);
  int steps = 0;
  // relative block norm of the last two visits of every position of the shift cycle
  std::vector<double> rel_h1(ns, 0.0), rel_h2(ns, 0.0);
  std::vector<double> ps_var, rinv_var, cinv_var, cinv_kept, rdummy, hn;
  for (int sw = 0;; ++sw) {
    // Width of this sweep.  With C = R^T R (R upper triangular) column block j of U R^-1 lies in
    // span{U_1..U_j}: it IS the block the step-by-step iteration appends at step j (up to its
    // sign), so the reference's stopping rule -- relative norm of the new block below
    // adi_newZ_reltol (optcont_main.py:123-124) -- is applied block by block below, and the
    // iteration ends after the same step as the sequential one.  So that the solves behind
    // the stopping step are not spent in vain, the block norms of the last two passes over
    // the shift cycle predict that step (per cycle position: same shift, geometric decay)
    // and the sweep is cut there (any run of consecutive, distinct shifts is a valid sweep;
    // its Cauchy data are computed on the spot).
    int g_now = G;
    if (prm.adi_newZ_reltol > 0.0) {
      for (int g = 0; g < G; ++g) {
        const int pos = (steps + g) % ns;
        if (rel_h1[pos] > 0.0 && rel_h2[pos] > rel_h1[pos]) {
          const double pred = rel_h1[pos] * (rel_h1[pos] / rel_h2[pos]);
          if (pred < prm.adi_newZ_reltol) {
            g_now = g + 1;
            break;
          }
        }
      }
    }
    g_now = std::min(g_now, prm.adi_max_steps - steps);
    if (g_now < 1) break;
    const std::vector<double>* psp;
    const std::vector<double>* rinvp;
    const std::vector<double>* cinvp;
    if (g_now == G && steps % G == 0) {
      psp = &pss[(steps / G) % ncyc];
      rinvp = &rinvs[(steps / G) % ncyc];
      cinvp = &cinvs[(steps / G) % ncyc];
    } else {
      ps_var.resize(g_now);
      for (int g = 0; g < g_now; ++g) ps_var[g] = shifts[(steps + g) % ns];
      rinv_var.assign((size_t)g_now * g_now, 0.0);
      cinv_var.assign(g_now, 0.0);
      if (cauchy_data(ps_var.data(), g_now, rinv_var.data(), cinv_var.data()) != RICADI_OK)
        throw HipError{"Cauchy matrix of a partial ADI sweep is numerically singular"};
      psp = &ps_var;
      rinvp = &rinv_var;
      cinvp = &cinv_var;
    }
    const std::vector<double>& ps = *psp;
    const std::vector<double>& rinv = *rinvp;
    const std::vector<double>& cinv1 = *cinvp;
    const int Gs = g_now;
    // who solves what, and where solution g sits in the buffer the recombination reads
    std::vector<int> slot_of(Gs), mine;
    int per_rank = Gs;
    if (shard) {
      std::vector<int> cnt(world, 0);
      for (int g = 0; g < Gs; ++g) {
        const int r = owner[(steps + g) % ns];
        slot_of[g] = cnt[r]++;                       // index among its owner's items, completed below
        if (r == rank) mine.push_back(g);
      }
      per_rank = *std::max_element(cnt.begin(), cnt.end());
      for (int g = 0; g < Gs; ++g) slot_of[g] += owner[(steps + g) % ns] * per_rank;
    } else {
      for (int g = 0; g < Gs; ++g) {
        slot_of[g] = g;
        mine.push_back(g);
      }
    }
    const int nslot = world * per_rank, nmine = (int)mine.size();
    std::vector<double> psm(nmine);
    for (int k = 0; k < nmine; ++k) psm[k] = ps[mine[k]];
    if (nmine && fail.empty()) guarded([&] { get_shifts(c, psm.data(), be.data(), nmine, sds.data()); });
    lap(c->t_setup);
    double* usolve = shard ? c->xsend : c->sweep_u.p;
    if (shard) {
      if ((size_t)per_rank * nm * sizeof(double) > exchange_panel_capacity(c))
        throw HipError{"exchange buffer too small: " + std::to_string((size_t)per_rank * nm * sizeof(double) + RICADI_XCTL) +
                       " bytes per rank needed, " + std::to_string(c->xcap) + " given to ricadi_set_exchange"};
      // padding slots travel as zeros (their coefficients are zero, but 0 * NaN is not)
      if (nmine < per_rank)
        HIPCHK(hipMemsetAsync(c->xsend + (size_t)nmine * nm, 0, sizeof(double) * nm * (per_rank - nmine), st));
    }
    if (nmine && fail.empty())
      guarded([&] {
        // test hook (tests/test_gpu_round4.py): this rank's share of sweep k fails
        if (const char* inj = shard ? getenv("RICADI_INJECT_SWEEP_FAILURE") : nullptr)
          if (atoi(inj) == sw) throw HipError{"injected failure in sweep " + std::to_string(sw)};
        load_rhs(c, dW, m, c->bvec.p);
        solve_batch(c, sds.data(), nmine, c->bvec.p, 0, usolve, m, true, nullptr, res.data());
      });
    c->lr_ucol = -1;            // only the first solve of a Newton step has U among its rhs columns
    lap(c->t_solve);
    if (fail.empty()) {
      for (int k = 0; k < nmine; ++k)
        if (!res[k].converged) {
          stt.nonconverged++;
          stt.worst_relres = std::max(stt.worst_relres, res[k].max_relres);
        }
      stt.shift_solves += nmine;
    }
    const double* ubase = usolve;
    double words[2] = {fail.empty() ? 0.0 : 1.0, 0.0};
    if (shard) {
      if (!fail.empty()) HIPCHK(hipMemsetAsync(c->xsend, 0, sizeof(double) * nm * per_rank, st));
      if (words_fit) {
        HIPCHK(hipMemcpyAsync(c->xsend + (size_t)nv * m, words, sizeof(words), hipMemcpyHostToDevice, st));
      } else {
        // no pressure rows to carry the words: a control message of their own
        double any = words[0];
        reduce_over_ranks(c, &any, 0, 1);
        if (any != 0.0)
          throw HipError{fail.empty() ? "another rank failed in its share of an ADI sweep" : fail};
      }
      exchange(c, (size_t)per_rank * nm);
      ubase = c->xrecv;
    }
    // coefficient rows (replicated over the m columns), in buffer order: Gs columns of R^-1, then C^-1 1
    coef.assign((size_t)(Gs + 1) * nslot * m, 0.0);
    auto fill_row = [&](int j, const double* col, int stride, int cnt) {   // row j <- col[i * stride], i < cnt
      for (int i = 0; i < cnt; ++i)
        for (int cidx = 0; cidx < m; ++cidx) coef[((size_t)j * nslot + slot_of[i]) * m + cidx] = col[(size_t)i * stride];
    };
    for (int j = 0; j < Gs; ++j) fill_row(j, rinv.data() + j, Gs, Gs);
    c->sweep_coef.ensure(coef.size());
    HIPCHK(hipMemcpyAsync(c->sweep_coef.p, coef.data(), sizeof(double) * (size_t)Gs * nslot * m,
                          hipMemcpyHostToDevice, st));
    // Z <- [Z, U R^-1]: block j = sum_i rinv[i][j] U_i, with its squared norm
    const bool combined = sweep_combine_ok(m, nslot, Gs);
    if (combined) {
      // all blocks and their norms in two launches (K4s)
      c->sweep_part.ensure(sweep_combine_partial_len(nv, m, Gs));
      launch_sweep_combine(st, nv, m, nslot, Gs, ubase, nm, c->sweep_coef.p, c->Z.p, c->zld, c->zc,
                           c->sweep_part.p, c->nrm2.p);
    } else {
      for (int j = 0; j < Gs; ++j) {
        launch_cols_update(st, nv, m, nslot, ubase, nm, c->sweep_coef.p + (size_t)j * nslot * m, 1.0,
                           nullptr, nullptr, c->sweep_t.p);
        launch_copy_cols(st, nv, m, c->sweep_t.p, m, 0, c->Z.p, c->zld, c->zc + j * m, 1.0);
        col_norms2(c, c->sweep_t.p, nv, m, c->nrm2.p + (size_t)j * m);
      }
    }
    hn.resize((size_t)Gs * m);
    HIPCHK(hipMemcpyAsync(hn.data(), c->nrm2.p, sizeof(double) * Gs * m, hipMemcpyDeviceToHost, st));
    std::vector<double> rwords;
    if (shard && words_fit) {
      // the ranks' status words, one strided copy out of the gathered buffer
      rwords.assign((size_t)2 * world, 0.0);
      HIPCHK(hipMemcpy2DAsync(rwords.data(), sizeof(double) * 2, c->xrecv + (size_t)nv * m,
                              sizeof(double) * nm * per_rank, sizeof(double) * 2, world, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    for (int r = 0; r < (int)rwords.size() / 2; ++r)
      if (rwords[(size_t)2 * r] != 0.0)
        throw HipError{r == rank && !fail.empty() ? fail
                                                  : "rank " + std::to_string(r) + " failed in its share of an ADI sweep"};
    // The block norms steer the stopping decisions, which must not differ between the ranks.  The fused
    // recombination sums in a fixed order (sweep_combine_kernel): every rank gets the same bits from the same
    // gathered panels and decides alone.  The per-block fallback hands round rank 0's values.
    if (!combined)
      for (int o = 0; o < Gs * m; o += RICADI_XCTL / 8)
        values_of_rank0(c, hn.data() + o, std::min(RICADI_XCTL / 8, Gs * m - o));
    // the reference's rule, block by block; blocks behind the stopping step are dropped
    int kept = Gs;
    bool stop = false;
    for (int j = 0; j < Gs; ++j) {
      double b2 = 0.0;
      for (int cc = 0; cc < m; ++cc) b2 += hn[(size_t)j * m + cc];
      znorm2 += b2;
      const double relj = znorm2 > 0.0 ? std::sqrt(b2 / znorm2) : 0.0;
      const int pos = (steps + j) % ns;
      rel_h2[pos] = rel_h1[pos];
      rel_h1[pos] = relj;
      stt.rel = relj;
      if (relj < prm.adi_newZ_reltol) {
        kept = j + 1;
        stop = true;
        break;
      }
    }
    // W <- W + E (U C^-1 1) over the blocks that are KEPT: every U_g was solved against the same W, so
    // the first `kept` solutions are the sweep of the first `kept` shifts, whose Cauchy data differ only
    // in C^-1 1 (R^-1 of the leading block is the leading block of R^-1) -- W stays the residual factor
    // of the truncated Z, and ||W^T W|| the residual norm that is reported
    const double* cw = cinv1.data();
    if (kept < Gs) {
      cinv_kept.assign(kept, 0.0);
      rdummy.assign((size_t)kept * kept, 0.0);
      if (cauchy_data(ps.data(), kept, rdummy.data(), cinv_kept.data()) != RICADI_OK)
        throw HipError{"Cauchy matrix of a truncated ADI sweep is numerically singular"};
      cw = cinv_kept.data();
    }
    fill_row(Gs, cw, 1, kept);
    HIPCHK(hipMemcpyAsync(c->sweep_coef.p + (size_t)Gs * nslot * m, coef.data() + (size_t)Gs * nslot * m,
                          sizeof(double) * (size_t)nslot * m, hipMemcpyHostToDevice, st));
    launch_cols_update(st, nv, m, nslot, ubase, nm, c->sweep_coef.p + (size_t)Gs * nslot * m, 1.0,
                       nullptr, nullptr, c->sweep_t.p);
    launch_spmm(st, nv, c->E.rp.p, c->E.ci.p, c->E.v.p, c->sweep_t.p, m, nullptr, dW, m, dW, m, 1.0,
                1.0, nullptr, m);
    HIPCHK(hipStreamSynchronize(st));     // `coef` is reused by the next sweep
    c->zc += kept * m;
    steps += kept;
    stt.steps = steps;
    stt.sweeps = sw + 1;
    lap(c->t_recomb);
    static const bool dbg = getenv("RICADI_DEBUG_SWEEPS") != nullptr;
    if (prm.verbose || dbg) {
      int its = 0;
      for (int k = 0; k < nmine; ++k) its = std::max(its, res[k].iters);
      if (dbg) {
        double wf = 0.0;
        DScalar::gram_norms(c, dW, c->nv, m, &wf, nullptr);
        fprintf(stderr, "[ricadi rank %d] sweep %d: Gs %d kept %d per_rank %d nmine %d  ||W^T W|| %.6e  znorm2 %.6e  its", rank, sw + 1,
                Gs, kept, per_rank, nmine, wf, znorm2);
        for (int k = 0; k < nmine; ++k) fprintf(stderr, " %d", res[k].iters);
        fprintf(stderr, "\n");
      }
      fprintf(stderr, "[ricadi] ADI sweep %3d (steps %d..%d): rel new Z %9.3e, gmres its <= %d%s\n",
              sw + 1, steps - kept + 1, steps, stt.rel, its, shard ? " (this rank)" : "");
    }
    if (stop) break;
    if (steps >= prm.adi_max_steps) break;
    if (prm.compress_cols > 0 && c->zc - zc_last >= prm.compress_cols) {
      // splice in what the helper finished during the last sweeps, hand it the next prefix
      job.finish();
      job.start();
      zc_last = c->zc;
      lap(c->t_compress);
    }
  }
  job.finish();
  lap(c->t_compress);
  stt.gmres_iters = c->total_iters - it0;
  if (shard) {
    // a rank has only seen its own solves
    double v[4] = {(double)stt.gmres_iters, (double)stt.shift_solves, (double)stt.nonconverged, stt.worst_relres};
    reduce_over_ranks(c, v, 3, 1);
    stt.gmres_iters = (long)(v[0] + 0.5);
    stt.shift_solves = (long)(v[1] + 0.5);
    stt.nonconverged = (long)(v[2] + 0.5);
    stt.worst_relres = v[3];
  }
  DScalar::gram_norms(c, dW, c->nv, m, &stt.res_fro, nullptr);
  return true;
}

// Depth of the recycling ring inside the ADI drivers (RICADI_RECYCLE=d; 0 switches it off)
// (cfg2, same-call A/B: depth 0 / 2 / 3 / 5 / 8 -> 63.1 / 56.3 / 55.3 / 53.4 / 52.7 iterations per solve,
// 436.7 / 405.9 / 405.9 / 401.9 / 409.8 ms per step.)  Every stored pair costs n x m doubles per shift:
// 5 where that is small, 3 beyond n = 2e5 (cfg5: 128 shifts x 3 x 64 MB).
static int adi_recycle_depth(const ricadi_ctx* c) {
  const char* e = getenv("RICADI_RECYCLE");     // read per call: tests toggle it
  return e ? std::max(0, std::min(8, atoi(e))) : (c->n <= 200000 ? 5 : 3);
}

static AdiStats lyap_adi_dev(ricadi_ctx* c, const double* shifts, int ns, double* dW, int m,
                             const ricadi_adi_params& prm) {
  AdiStats stt;
  Restore<int> keep_rec(c->rec_depth);
  c->rec_depth = std::max(c->rec_user_depth, adi_recycle_depth(c));
  if (prm.sweep_width > 1 && lyap_adi_sweeps_dev(c, shifts, ns, dW, m, prm, stt)) return stt;
  stt = AdiStats();
  hipStream_t st = c->st;
  ensure_work(c, m);
  // per-shift data of the whole shift cycle (and of the projection) up front: the coarse
  // inverses then come out of one batched factorisation instead of one at a time
  prefetch_setup(c, shifts, std::min(ns, prm.adi_max_steps), prm.project_w != 0);
  if (prm.project_w) project_panel(c, dW, m);
  const long it0 = c->total_iters;
  double znorm2 = 0.0;
  int zc_last = c->zc;
  for (int step = 1; step <= prm.adi_max_steps; ++step) {
    const double p = shifts[(step - 1) % ns];
    ShiftData* sd = get_shift(c, p, 1.0);
    load_rhs(c, dW, m, c->bvec.p);
    GmresResult r = gmres_solve(c, sd, c->bvec.p, c->xs.p, m, true, nullptr);
    if (!r.converged) {
      stt.nonconverged++;
      stt.worst_relres = std::max(stt.worst_relres, r.max_relres);
      if (prm.verbose)
        fprintf(stderr, "[ricadi] ADI step %d shift %g: GMRES stopped at relres %.2e after %d its\n",
                step, p, r.max_relres, r.iters);
    }
    stt.shift_solves++;
    // W <- W - 2 p E V
    launch_spmm(st, c->nv, c->E.rp.p, c->E.ci.p, c->E.v.p, c->xs.p, m, nullptr, dW, m, dW, m,
                -2.0 * p, 1.0, nullptr, m);
    // Z <- [Z, sqrt(-2p) V]
    launch_copy_cols(st, c->nv, m, c->xs.p, m, 0, c->Z.p, c->zld, c->zc, std::sqrt(-2.0 * p));
    double n2 = 0.0;
    col_norms2(c, c->xs.p, c->nv, m, c->nrm2.p);
    HIPCHK(hipMemcpyAsync(c->h_resid, c->nrm2.p, sizeof(double) * m, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int j = 0; j < m; ++j) n2 += c->h_resid[j];
    n2 *= -2.0 * p;
    znorm2 += n2;
    c->zc += m;
    stt.steps = step;
    stt.rel = znorm2 > 0.0 ? std::sqrt(n2 / znorm2) : 0.0;
    if (prm.verbose)
      fprintf(stderr, "[ricadi] ADI step %3d: shift %10.3e rel new Z %9.3e gmres its %d\n", step,
              p, stt.rel, r.iters);
    if (stt.rel < prm.adi_newZ_reltol) break;
    if (prm.compress_cols > 0 && c->zc - zc_last >= prm.compress_cols) {
      factor_recompress(c);
      zc_last = c->zc;
    }
  }
  stt.gmres_iters = c->total_iters - it0;
  DScalar::gram_norms(c, dW, c->nv, m, &stt.res_fro, nullptr);
  return stt;
}

static void factor_reserve(ricadi_ctx* c, int ld) {
  if ((size_t)c->nv * ld > c->Z.n) c->Z.alloc((size_t)c->nv * ld);
  c->zld = ld;
  c->zc = 0;
}

