// solver_capi.inl -- the C-ABI of include/ricadi.h.
// Part of ricadi_solver.hip (one translation unit; included there in order).

// =====================================================================================
//                                      C  A B I
// =====================================================================================
#define API_BEGIN try {
#define API_END                                                   \
  }                                                               \
  catch (const ricadi::HipError& e) {                             \
    ricadi::set_error(e.msg);                                     \
    return RICADI_EHIP;                                           \
  }                                                               \
  catch (const std::exception& e) {                               \
    ricadi::set_error(e.what());                                  \
    return RICADI_EHIP;                                           \
  }                                                               \
  catch (...) {                                                   \
    ricadi::set_error("unknown C++ exception");                   \
    return RICADI_EHIP;                                           \
  }                                                               \
  return RICADI_OK;

#define REQUIRE(cond, code, msg)     \
  do {                               \
    if (!(cond)) {                   \
      ricadi::set_error(msg);        \
      return code;                   \
    }                                \
  } while (0)

extern "C" {

const char* ricadi_last_error(void) { return ricadi::g_err.c_str(); }
int ricadi_version(void) { return 401; }
int ricadi_sizeof_opts(void) { return (int)sizeof(ricadi_opts); }
int ricadi_sizeof_adi_params(void) { return (int)sizeof(ricadi_adi_params); }
// field types in declaration order (d = double, i = int); keep in step with include/ricadi.h
const char* ricadi_struct_signature(void) { return "ricadi_opts:diiiiiiiiii;ricadi_adi_params:ididdiiii"; }

void ricadi_default_opts(ricadi_opts* o) {
  if (!o) return;
  o->gmres_tol = 1e-10;
  o->gmres_restart = 30;
  o->gmres_maxit = 3000;
  o->bj_block = 32;
  o->agg_v = 16;
  o->agg_p = 24;
  o->coarse_max = 4096;
  o->use_coarse = 1;
  o->max_levels = 3;
  o->verbose = 0;
  o->compress_qr = 1;
}

void ricadi_default_adi_params(ricadi_adi_params* p) {
  if (!p) return;
  // /root/reference/optcont_main.py:122-131
  p->adi_max_steps = 200;
  p->adi_newZ_reltol = 1e-8;
  p->nwtn_max_steps = 16;
  p->nwtn_upd_reltol = 5e-8;
  p->nwtn_upd_abstol = 1e-7;
  p->project_w = 1;
  p->verbose = 0;
  p->compress_cols = 0;
  p->sweep_width = 1;
}

int ricadi_create(int device_id, ricadi_ctx** out) {
  REQUIRE(out, RICADI_EINVAL, "ricadi_create: ctx is NULL");
  *out = nullptr;
  API_BEGIN
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) throw ricadi::HipError{"no HIP device visible (this library has no CPU fallback)"};
  if (device_id < 0 || device_id >= ndev) throw ricadi::HipError{"bad device id"};
  HIPCHK(hipSetDevice(device_id));
  std::unique_ptr<ricadi_ctx> c(new ricadi_ctx);
  c->dev = device_id;
  ricadi_default_opts(&c->opts);
  c->precond32 = getenv("RICADI_PRECOND64") == nullptr;
  c->timing = getenv("RICADI_TIMING") != nullptr;
  if (const char* e = getenv("RICADI_SMW")) c->smw = e[0] != '0';
  HIPCHK(hipStreamCreate(&c->st));
  RBCHK(rocblas_create_handle(&c->rb));
  RBCHK(rocblas_set_stream(c->rb, c->st));
  c->flag.alloc(4);
  c->info.alloc(4);
  *out = c.release();
  API_END
}

int ricadi_destroy(ricadi_ctx* ctx) {
  if (!ctx) return RICADI_OK;
  API_BEGIN
  (void)hipSetDevice(ctx->dev);
  (void)hipStreamSynchronize(ctx->st);
  delete ctx;
  API_END
}

int ricadi_set_opts(ricadi_ctx* c, const ricadi_opts* o) {
  REQUIRE(c && o, RICADI_EINVAL, "ricadi_set_opts: NULL argument");
  REQUIRE(o->gmres_restart >= 2 && o->gmres_restart <= 400, RICADI_EINVAL, "gmres_restart out of range");
  REQUIRE(o->gmres_tol > 0 && o->gmres_maxit > 0, RICADI_EINVAL, "bad gmres_tol / gmres_maxit");
  const bool structural = c->has_op && (o->bj_block != c->opts.bj_block || o->agg_v != c->opts.agg_v ||
                                        o->agg_p != c->opts.agg_p || o->coarse_max != c->opts.coarse_max ||
                                        o->max_levels != c->opts.max_levels ||
                                        o->use_coarse != c->opts.use_coarse);
  REQUIRE(!structural, RICADI_ESTATE, "preconditioner options must be set before ricadi_set_operator");
  c->opts = *o;
  return RICADI_OK;
}

void* ricadi_stream(ricadi_ctx* c) { return c ? (void*)c->st : nullptr; }

int ricadi_synchronize(ricadi_ctx* c) {
  REQUIRE(c, RICADI_EINVAL, "NULL ctx");
  API_BEGIN
  HIPCHK(hipStreamSynchronize(c->st));
  API_END
}

int ricadi_set_operator(ricadi_ctx* c, int nv, int np, const int32_t* a_rp, const int32_t* a_ci,
                        const double* a_v, const int32_t* e_rp, const int32_t* e_ci,
                        const double* e_v, const int32_t* j_rp, const int32_t* j_ci,
                        const double* j_v) {
  REQUIRE(c, RICADI_EINVAL, "NULL ctx");
  REQUIRE(nv > 0 && np >= 0, RICADI_EINVAL, "bad sizes");
  REQUIRE(a_rp && e_rp, RICADI_EINVAL, "NULL matrix");
  REQUIRE((a_ci && a_v) || a_rp[nv] == 0, RICADI_EINVAL, "NULL matrix arrays");
  REQUIRE((e_ci && e_v) || e_rp[nv] == 0, RICADI_EINVAL, "NULL matrix arrays");
  REQUIRE(np == 0 || (j_rp && j_ci && j_v), RICADI_EINVAL, "NULL J");
  API_BEGIN
  HIPCHK(hipSetDevice(c->dev));
  hipStream_t st = c->st;
  HostCsr A = make_csr(nv, nv, a_rp, a_ci, a_v);
  HostCsr E = make_csr(nv, nv, e_rp, e_ci, e_v);
  HostCsr J;
  if (np > 0) {
    J = make_csr(np, nv, j_rp, j_ci, j_v);
  } else {
    J.nrows = 0;
    J.ncols = nv;
    J.rp.assign(1, 0);
  }
  for (size_t k = 0; k < A.nnz(); ++k)
    if (A.ci[k] < 0 || A.ci[k] >= nv) throw ricadi::HipError{"A: column index out of range"};
  for (size_t k = 0; k < E.nnz(); ++k)
    if (E.ci[k] < 0 || E.ci[k] >= nv) throw ricadi::HipError{"E: column index out of range"};
  for (size_t k = 0; k < J.nnz(); ++k)
    if (J.ci[k] < 0 || J.ci[k] >= nv) throw ricadi::HipError{"J: column index out of range"};
  HostSetup hs;
  if (!c->borrowed)
    c->levels = std::max(2, c->opts.max_levels);
  // smoothed aggregation of the velocity prolongation (two-level setups, folded preconditioner cycle only);
  // RICADI_SA=0 switches it off, RICADI_SA=<omega> sets the damping
  double sa_omega = getenv("RICADI_SA") ? atof(getenv("RICADI_SA")) : 0.5;
  if (c->borrowed || np == 0 || c->opts.bj_block != 32) sa_omega = 0.0;
  build_setup(A, E, J, c->opts, hs, c->levels, sa_omega);
  if (hs.sa) {
    // the folded first sweep takes per-block dense slices of S*P of at most 64 columns
    int kmax = 0;
    std::vector<int> tmp;
    for (int b = 0; b < hs.nbv; ++b) {
      tmp.clear();
      for (int q = hs.bv_ptr[b]; q < hs.bv_ptr[b + 1]; ++q)
        for (int kk = hs.sy_rp[hs.bv_rows[q]]; kk < hs.sy_rp[hs.bv_rows[q] + 1]; ++kk) tmp.push_back(hs.sy_ci[kk]);
      std::sort(tmp.begin(), tmp.end());
      kmax = std::max(kmax, (int)(std::unique(tmp.begin(), tmp.end()) - tmp.begin()));
    }
    if (kmax > 64 || !block_apply2_ok(hs.bs, 64)) {
      if (c->opts.verbose)
        fprintf(stderr, "[ricadi] smoothed aggregation off: a velocity block touches %d coarse columns\n", kmax);
      hs = HostSetup();
      build_setup(A, E, J, c->opts, hs, c->levels, 0.0);
    }
  }
  c->sa = hs.sa;
  c->cache.clear();
  c->child.reset();
  if (hs.multilevel) {
    std::unique_ptr<ricadi_ctx> ch(new ricadi_ctx);
    ch->dev = c->dev;
    ch->st = c->st;
    ch->rb = c->rb;
    ch->borrowed = true;
    ch->opts = c->opts;
    // aggregates of the child level (in units of ITS dofs = this level's aggregates); they double
    // until the last level's dense inverse fits coarse_max
    ch->opts.agg_v = 2;
    ch->opts.agg_p = 1;
    ch->opts.coarse_max = c->opts.coarse_max + c->opts.coarse_max / 8;   // pairs do not always pair up
    ch->levels = 2;
    ch->precond32 = c->precond32;
    ch->smw = c->smw;
    ch->flag.alloc(4);
    ch->info.alloc(4);
    const int rc = ricadi_set_operator(ch.get(), hs.kcv, hs.kcp, hs.l1A.rp.data(), hs.l1A.ci.data(), hs.l1A.v.data(),
                                       hs.l1E.rp.data(), hs.l1E.ci.data(), hs.l1E.v.data(), hs.l1J.rp.data(),
                                       hs.l1J.ci.data(), hs.l1J.v.data());
    if (rc != RICADI_OK) throw ricadi::HipError{std::string("child level: ") + ricadi_last_error()};
    c->child = std::move(ch);
  }
  c->nv = nv;
  c->np = np;
  c->n = nv + np;
  c->bs = hs.bs;
  c->nbv = hs.nbv;
  c->nbp = hs.nbp;
  c->kc = hs.kc;
  c->snnz = hs.s_ci.size();
  c->s_rp.upload(hs.s_rp, st);
  c->s_ci.upload(hs.s_ci, st);
  c->srcA.upload(hs.s_srcA, st);
  c->srcE.upload(hs.s_srcE, st);
  c->srcJ.upload(hs.s_srcJ, st);
  c->A.upload(A, st);
  c->E.upload(E, st);
  c->J.upload(J, st);
  HostCsr JT = transpose(J);
  c->JT.upload(JT, st);
  std::vector<int> sw_gptr, sw_gcols, sw_cptr, sw_ccols;      // host copies for the sweeps' fixed-stride records
  {
    // rectangular last sweep: pressure dofs touched by every velocity block, dense J^T slices
    c->gt_ok = false;
    if (np > 0 && hs.nbv > 0) {
      std::vector<int> gptr(hs.nbv + 1, 0), gcols;
      int kmax = 0;
      std::vector<int> tmp;
      for (int b = 0; b < hs.nbv; ++b) {
        tmp.clear();
        for (int q = hs.bv_ptr[b]; q < hs.bv_ptr[b + 1]; ++q) {
          const int row = hs.bv_rows[q];
          for (int k = JT.rp[row]; k < JT.rp[row + 1]; ++k) tmp.push_back(JT.ci[k]);
        }
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        gcols.insert(gcols.end(), tmp.begin(), tmp.end());
        gptr[b + 1] = (int)gcols.size();
        kmax = std::max(kmax, (int)tmp.size());
      }
      const int ks = kmax <= 32 ? 32 : (kmax <= 64 ? 64 : (kmax <= 128 ? 128 : 0));
      if (ks > 0 && block_apply_rect_ok(hs.bs, ks)) {
        std::vector<double> jtd((size_t)hs.nbv * hs.bs * ks, 0.0);
        for (int b = 0; b < hs.nbv; ++b) {
          const int* cb = gcols.data() + gptr[b];
          const int nc = gptr[b + 1] - gptr[b];
          for (int q = hs.bv_ptr[b]; q < hs.bv_ptr[b + 1]; ++q) {
            const int row = hs.bv_rows[q], il = q - hs.bv_ptr[b];
            for (int k = JT.rp[row]; k < JT.rp[row + 1]; ++k) {
              const int jl = (int)(std::lower_bound(cb, cb + nc, JT.ci[k]) - cb);
              jtd[((size_t)b * hs.bs + il) * ks + jl] += JT.v[k];
            }
          }
        }
        c->gt_ptr.upload(gptr, st);
        c->gt_cols.upload(gcols, st);
        sw_gptr = gptr;
        sw_gcols = gcols;
        c->gt_jtd.upload(jtd, st);
        c->gt_ks = ks;
        c->gt_ok = true;
        if (c->opts.verbose)
          fprintf(stderr, "[ricadi] last velocity sweep in rectangular form: <= %d pressure dofs per block (slice width %d)\n",
                  kmax, ks);
      }
    }
  }
  c->bv_ptr.upload(hs.bv_ptr, st);
  c->bv_rows.upload(hs.bv_rows, st);
  c->bp_ptr.upload(hs.bp_ptr, st);
  c->bp_rows.upload(hs.bp_rows, st);
  c->bvA.upload(hs.bv_A, st);
  c->bvE.upload(hs.bv_E, st);
  c->jd_ptr.upload(hs.jd_ptr, st);
  c->jd_vblk.upload(hs.jd_vblk, st);
  c->jd_val.upload(hs.jd_val, st);
  c->agg_ptr.upload(hs.agg_ptr, st);
  c->agg_rows.upload(hs.agg_rows, st);
  c->aggof.upload(hs.aggof, st);
  if (hs.sa) {
    c->pt_rp.upload(hs.pt_rp, st);
    c->pt_ci.upload(hs.pt_ci, st);
    c->pt_v.upload(hs.pt_v, st);
  }
  c->synnz = hs.sy_ci.size();
  c->sy_chunk = (c->synnz <= (size_t)10 * std::max(c->n, 1)) ? 8 : 16;
  if (c->opts.verbose)
    fprintf(stderr, "[ricadi] prolongated operator S*Y: %.1f entries per row\n",
            (double)c->synnz / std::max(c->n, 1));
  c->sy_rp.upload(hs.sy_rp, st);
  c->sy_ci.upload(hs.sy_ci, st);
  {
    // records of the fused pressure step (pressure_step_kernel): one load per (block, row) instead of the chain
    // block list -> row index -> row pointers
    std::vector<int> meta((size_t)std::max(hs.nbp, 0) * 32 * 5, 0);
    const bool with_sy = hs.kc > 0 && (int)hs.sy_rp.size() == hs.n + 1;
    for (int b = 0; b < hs.nbp; ++b)
      for (int il = 0; il < 32; ++il) {
        int* mt = &meta[((size_t)b * 32 + il) * 5];
        const int cnt = hs.bp_ptr[b + 1] - hs.bp_ptr[b];
        if (il >= cnt || cnt > 32) {
          mt[0] = -1;
          continue;
        }
        const int prow = hs.bp_rows[hs.bp_ptr[b] + il];
        mt[0] = prow;
        mt[1] = J.rp[prow];
        mt[2] = J.rp[prow + 1];
        mt[3] = with_sy ? hs.sy_rp[hs.nv + prow] : 0;
        mt[4] = with_sy ? hs.sy_rp[hs.nv + prow + 1] : 0;
        // (the kernel clamps its index loads to the row's last entry: an empty row must not point behind the arrays)
        if (mt[2] == mt[1]) mt[1] = mt[2] = 0;
        if (mt[4] == mt[3]) mt[3] = mt[4] = 0;
      }
    c->ps_meta.upload(meta, st);
  }
  c->sy_A.upload(hs.sy_A, st);
  c->sy_E.upload(hs.sy_E, st);
  c->sy_J.upload(hs.sy_J, st);
  {
    // dense slices of S*Y per velocity block (first sweep with the coarse residual folded in)
    c->ady_ok = false;
    if (hs.kc > 0 && np > 0 && hs.nbv > 0 && !hs.sy_rp.empty()) {
      std::vector<int> cptr(hs.nbv + 1, 0), ccols, tmp;
      int kmax = 0;
      for (int b = 0; b < hs.nbv; ++b) {
        tmp.clear();
        for (int q = hs.bv_ptr[b]; q < hs.bv_ptr[b + 1]; ++q) {
          const int row = hs.bv_rows[q];
          for (int kk = hs.sy_rp[row]; kk < hs.sy_rp[row + 1]; ++kk) tmp.push_back(hs.sy_ci[kk]);
        }
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        ccols.insert(ccols.end(), tmp.begin(), tmp.end());
        cptr[b + 1] = (int)ccols.size();
        kmax = std::max(kmax, (int)tmp.size());
      }
      const int ks = kmax <= 32 ? 32 : (kmax <= 64 ? 64 : 0);
      if (ks > 0 && block_apply2_ok(hs.bs, ks)) {
        const size_t tot = (size_t)hs.nbv * hs.bs * ks;
        std::vector<double> dA(tot, 0.0), dE(tot, 0.0), dJ(tot, 0.0), dT(hs.sa ? tot : 0, 0.0);
        for (int b = 0; b < hs.nbv; ++b) {
          const int* cb = ccols.data() + cptr[b];
          const int nc = cptr[b + 1] - cptr[b];
          for (int q = hs.bv_ptr[b]; q < hs.bv_ptr[b + 1]; ++q) {
            const int row = hs.bv_rows[q], il = q - hs.bv_ptr[b];
            for (int kk = hs.sy_rp[row]; kk < hs.sy_rp[row + 1]; ++kk) {
              const int jl = (int)(std::lower_bound(cb, cb + nc, hs.sy_ci[kk]) - cb);
              const size_t at = ((size_t)b * hs.bs + il) * ks + jl;
              dA[at] += hs.sy_A[kk];
              dE[at] += hs.sy_E[kk];
              dJ[at] += hs.sy_J[kk];
            }
            if (hs.sa)       // (P - Y)[row, :]: its columns are among those of (S P)[row, :] (S has a diagonal)
              for (int kk = hs.pd_rp[row]; kk < hs.pd_rp[row + 1]; ++kk) {
                const int* f = std::lower_bound(cb, cb + nc, hs.pd_ci[kk]);
                if (f == cb + nc || *f != hs.pd_ci[kk]) throw ricadi::HipError{"smoothed prolongation: column outside the block's list"};
                dT[((size_t)b * hs.bs + il) * ks + (int)(f - cb)] += hs.pd_v[kk];
              }
          }
        }
        if (hs.sa) c->cy_dT.upload(dT, st);
        c->cy_ptr.upload(cptr, st);
        c->cy_cols.upload(ccols, st);
        sw_cptr = cptr;
        sw_ccols = ccols;
        c->cy_dA.upload(dA, st);
        c->cy_dE.upload(dE, st);
        c->cy_dJ.upload(dJ, st);
        c->ady_ks = ks;
        c->ady_ok = true;
      }
    }
  }
  {
    // fixed-stride records of the velocity sweeps (block_apply2_kernel, block_apply_rect_kernel; ProlongArgs::bmeta)
    c->sw_stride = 0;
    if (hs.bs == 32 && hs.nbv > 0 && (c->gt_ok || c->ady_ok)) {
      const int kr = c->gt_ok ? c->gt_ks : 0, k2 = c->ady_ok ? c->ady_ks : 0;
      const int stride = 68 + kr + k2;
      const std::vector<int>&gptr = sw_gptr, &gcols = sw_gcols, &cptr = sw_cptr, &ccols = sw_ccols;
      std::vector<int> meta((size_t)hs.nbv * stride, 0);
      bool ok = true;
      for (int b = 0; b < hs.nbv && ok; ++b) {
        int* mt = &meta[(size_t)b * stride];
        const int b0 = hs.bv_ptr[b], nb = hs.bv_ptr[b + 1] - b0;
        if (nb > 32 || nb <= 0) { ok = false; break; }
        mt[0] = nb;
        for (int i = 0; i < 32; ++i) {
          const int row = hs.bv_rows[b0 + std::min(i, nb - 1)];
          mt[4 + i] = row;
          mt[36 + i] = hs.kc > 0 ? hs.aggof[row] : 0;
        }
        if (c->gt_ok) {
          const int i0 = gptr[b], ni = gptr[b + 1] - i0;
          mt[1] = ni;
          for (int i = 0; i < kr; ++i) mt[68 + i] = ni > 0 ? gcols[i0 + std::min(i, ni - 1)] : 0;
        }
        if (c->ady_ok) {
          const int i0 = cptr[b], ni = cptr[b + 1] - i0;
          mt[2] = ni;
          for (int i = 0; i < k2; ++i) mt[68 + kr + i] = ni > 0 ? ccols[i0 + std::min(i, ni - 1)] : 0;
        }
      }
      const char* swe = getenv("RICADI_SWEEP_META");        // =0: the generic sweep kernels
      if (ok && !(swe && swe[0] == '0')) {
        c->sw_meta.upload(meta, st);
        c->sw_stride = stride;
        c->sw_in_rect = 68;
        c->sw_in_two = 68 + kr;
      }
    }
  }
  c->syb_ok = hs.kc > 0 && hs.sb_nblk > 0 && hs.syb_max_cols > 0;
  c->syb_max_cols = hs.syb_max_cols;
  if (c->syb_ok) {
    const int nb = hs.sb_nblk, mc = hs.syb_max_cols;
    std::vector<int> rp2((size_t)nb * 33, 0), cols2((size_t)nb * mc, -1);
    for (int b = 0; b < nb; ++b) {
      const int q0 = hs.sb_rowptr[b], nr = hs.sb_rowptr[b + 1] - q0;
      for (int q = 0; q <= 32; ++q) rp2[(size_t)b * 33 + q] = hs.syb_rp[q0 + std::min(q, nr)];
      const int c0 = hs.syb_cptr[b], nc = hs.syb_cptr[b + 1] - c0;
      for (int j = 0; j < nc; ++j) cols2[(size_t)b * mc + j] = hs.syb_cols[c0 + j];
    }
    c->syb_rp2.upload(rp2, st);
    c->syb_cols2.upload(cols2, st);
    c->syb_perm.upload(hs.syb_perm, st);
    c->syb_lidx.upload(hs.syb_lidx, st);
  }
  c->E0.upload(hs.E0, st);
  c->EM.upload(hs.EM, st);
  c->EJ.upload(hs.EJ, st);
  c->ones.upload(std::vector<double>((size_t)c->n, 1.0), st);
  c->sb_nblk = hs.sb_nblk;
  c->sb_max_cols = hs.sb_max_cols;
  c->sb_max_nnz = hs.sb_max_nnz;
  {
    const int nb = hs.sb_nblk, mc = std::max(hs.sb_max_cols, 1);
    std::vector<int> rows2((size_t)nb * 32, -1), rp2((size_t)nb * 33, 0), cols2((size_t)nb * mc, -1),
        colsm2((size_t)nb * mc, -1);
    for (int b = 0; b < nb; ++b) {
      const int q0 = hs.sb_rowptr[b], nr = hs.sb_rowptr[b + 1] - q0;
      for (int q = 0; q <= 32; ++q) rp2[(size_t)b * 33 + q] = hs.sb_rp[q0 + std::min(q, nr)];
      for (int q = 0; q < nr; ++q) rows2[(size_t)b * 32 + q] = hs.sb_rows[q0 + q];
      const int c0 = hs.sb_cptr[b], nc = hs.sb_cptr[b + 1] - c0;
      for (int j = 0; j < nc; ++j) {
        cols2[(size_t)b * mc + j] = hs.sb_cols[c0 + j];
        colsm2[(size_t)b * mc + j] = hs.kc > 0 ? hs.aggof[hs.sb_cols[c0 + j]] : -1;
      }
    }
    c->sb_rows2.upload(rows2, st);
    c->sb_rp2.upload(rp2, st);
    c->sb_cols2.upload(cols2, st);
    c->sb_colsm2.upload(colsm2, st);
  }
  c->sb_perm.upload(hs.sb_perm, st);
  c->sb_lidx.upload(hs.sb_lidx, st);
  c->sb_ok = hs.sb_nblk > 0 && hs.sb_max_cols < 65536;
  if (const char* e = getenv("RICADI_MS_SPMM")) {
    c->ms_spmm = e[0] != '0';
    c->ms_force = e[0] == '2';
  }
  if (const char* e = getenv("RICADI_MID32")) c->mid32 = e[0] != '0';
  if (const char* e = getenv("RICADI_ROWWAVE")) c->rowwave = e[0] != '0';
  if (const char* e = getenv("RICADI_COARSE32")) c->coarse_mfma32 = e[0] == '1';
  if (const char* e = getenv("RICADI_SWEEP32")) c->sweep_mfma32 = e[0] == '1';
  if (const char* e = getenv("RICADI_BLOCKS16")) c->blocks16 = e[0] != '0';
  if (const char* e = getenv("RICADI_X32")) c->x32_always = e[0] != '0';
  if (const char* e = getenv("RICADI_W32")) c->w32 = e[0] != '0';
  // multi-shift kernel operands: vAJ = A part + J part (disjoint supports) and vE in tile
  // order; velocity-velocity flag in bit 15 of the local index
  auto ms_arrays = [&](const std::vector<int>& rp, const std::vector<int>& ci, const std::vector<double>& a,
                       const std::vector<double>& e, const std::vector<double>& j, const std::vector<int>& perm,
                       const std::vector<uint16_t>& lidx, int ncol_v, DArr<double>& dAJ, DArr<double>& dE,
                       DArr<uint16_t>& dl) {
    const size_t nnz = perm.size();
    std::vector<int> rowof(ci.size());
    for (int i = 0; i + 1 < (int)rp.size(); ++i)
      for (int k = rp[i]; k < rp[i + 1]; ++k) rowof[k] = i;
    std::vector<double> aj(nnz), ee(nnz);
    std::vector<uint16_t> lm(nnz);
    for (size_t kb = 0; kb < nnz; ++kb) {
      const int k = perm[kb];
      aj[kb] = a[k] + j[k];
      ee[kb] = e[k];
      const bool vv = rowof[k] < nv && ci[k] < ncol_v;
      lm[kb] = (uint16_t)(lidx[kb] | (vv ? 0x8000 : 0));
    }
    dAJ.upload(aj, st);
    dE.upload(ee, st);
    dl.upload(lm, st);
  };
  if (c->sb_ok && hs.sb_max_cols <= 160)
    ms_arrays(hs.s_rp, hs.s_ci, hs.s_srcA, hs.s_srcE, hs.s_srcJ, hs.sb_perm, hs.sb_lidx, nv, c->sbAJ, c->sbE,
              c->sb_lidx_ms);
  else
    c->ms_spmm = false;
  if (c->syb_ok && hs.syb_max_cols <= 160 && c->ms_spmm)
    ms_arrays(hs.sy_rp, hs.sy_ci, hs.sy_A, hs.sy_E, hs.sy_J, hs.syb_perm, hs.syb_lidx, hs.kcv, c->sybAJ, c->sybE,
              c->syb_lidx_ms);
  HIPCHK(hipStreamSynchronize(st));
  c->q = 0;
  c->wcols = 0;  // workspaces depend on n
  c->zc = 0;
  c->has_op = true;
  if (c->opts.verbose)
    fprintf(stderr, "[ricadi] operator nv=%d np=%d nnz(S)=%zu | BJ blocks %d+%d (bs=%d) | coarse %d (%d+%d) | "
            "SpMM row blocks %d (max %d distinct cols, %d nnz; mean %.0f cols)\n",
            nv, np, c->snnz, c->nbv, c->nbp, c->bs, c->kc, hs.kcv, hs.kcp, hs.sb_nblk,
            hs.sb_max_cols, hs.sb_max_nnz, hs.sb_nblk ? (double)hs.sb_cols.size() / hs.sb_nblk : 0.0);
  API_END
}

int ricadi_clear_cache(ricadi_ctx* c) {
  REQUIRE(c, RICADI_EINVAL, "NULL ctx");
  API_BEGIN
  HIPCHK(hipStreamSynchronize(c->st));
  for (ricadi_ctx* l = c; l; l = l->child.get())
    for (auto& kv : l->cache) {
      kv.second->valid = false;   // buffers stay
      for (auto& r : kv.second->rec) r->serial = -1;
    }
  for (auto& e : c->rec_ring) e->serial = -1;
  API_END
}

int ricadi_set_recycle(ricadi_ctx* c, int depth) {
  REQUIRE(c && depth >= 0 && depth <= 8, RICADI_EINVAL, "recycling depth must be in [0, 8]");
  c->rec_user_depth = c->rec_depth = depth;
  return RICADI_OK;
}

// the exchange state of a context back to "none" (a communicator the library created is destroyed)
static void exchange_reset(ricadi_ctx* c) {
  if (c->xcomm && c->xcomm_owned) (void)ncclCommDestroy(c->xcomm);
  c->xcomm = nullptr;
  c->xcomm_owned = false;
  c->xforce = false;
  c->xsend_own.release();
  c->xrecv_own.release();
  c->xrank = 0;
  c->xworld = 1;
  c->xfn = nullptr;
  c->xuser = nullptr;
  c->xsend = c->xrecv = nullptr;
  c->xcap = 0;
}

int ricadi_set_exchange(ricadi_ctx* c, int rank, int world, ricadi_allgather_fn fn, void* user,
                        void* send_dev, void* recv_dev, int64_t send_capacity) {
  REQUIRE(c, RICADI_EINVAL, "NULL ctx");
  API_BEGIN
  exchange_reset(c);
  if (world <= 1 || !fn) return RICADI_OK;
  REQUIRE(rank >= 0 && rank < world && world <= 64, RICADI_EINVAL, "bad rank / world size");
  REQUIRE(send_dev && recv_dev && send_capacity >= 2 * RICADI_XCTL, RICADI_EINVAL, "exchange buffers missing or too small");
  c->xrank = rank;
  c->xworld = world;
  c->xfn = fn;
  c->xuser = user;
  c->xsend = static_cast<double*>(send_dev);
  c->xrecv = static_cast<double*>(recv_dev);
  c->xcap = (size_t)send_capacity;
  API_END
}

int ricadi_rccl_unique_id(void* id_out, int bytes) {
  REQUIRE(id_out && bytes >= (int)sizeof(ncclUniqueId), RICADI_EINVAL, "id buffer of at least 128 bytes required");
  ncclUniqueId id;
  const ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) {
    ricadi::set_error(std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
    return RICADI_EHIP;
  }
  std::memcpy(id_out, &id, sizeof(id));
  return RICADI_OK;
}

int ricadi_set_exchange_rccl(ricadi_ctx* c, int rank, int world, const void* unique_id, void* comm,
                             int64_t send_capacity) {
  REQUIRE(c, RICADI_EINVAL, "NULL ctx");
  REQUIRE(world >= 1 && world <= 64 && rank >= 0 && rank < world, RICADI_EINVAL, "bad rank / world size");
  const bool resize = !unique_id && !comm;      // keep the communicator, new buffer sizes
  REQUIRE(!resize || (c->xcomm && c->xrank == rank && c->xworld == world), RICADI_EINVAL,
          "a unique id (ricadi_rccl_unique_id) or a communicator is required");
  REQUIRE(send_capacity >= 2 * RICADI_XCTL, RICADI_EINVAL, "send_capacity too small");
  API_BEGIN
  HIPCHK(hipSetDevice(c->dev));
  if (resize) {
    HIPCHK(hipStreamSynchronize(c->st));
  } else if (comm) {
    exchange_reset(c);
    c->xcomm = static_cast<ncclComm_t>(comm);
  } else {
    exchange_reset(c);
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    const ncclResult_t r = ncclCommInitRank(&c->xcomm, world, id, rank);
    if (r != ncclSuccess) {
      c->xcomm = nullptr;
      throw HipError{std::string("ncclCommInitRank: ") + ncclGetErrorString(r)};
    }
    c->xcomm_owned = true;
  }
  const size_t cap = ((size_t)send_capacity + 7) / 8;
  c->xsend_own.alloc(cap);
  c->xrecv_own.alloc(cap * world);
  HIPCHK(hipMemsetAsync(c->xsend_own.p, 0, cap * sizeof(double), c->st));
  HIPCHK(hipMemsetAsync(c->xrecv_own.p, 0, cap * world * sizeof(double), c->st));
  c->xrank = rank;
  c->xworld = world;
  c->xforce = world == 1;
  c->xsend = c->xsend_own.p;
  c->xrecv = c->xrecv_own.p;
  c->xcap = cap * sizeof(double);
  API_END
}

int ricadi_exchange_count(ricadi_ctx* c, int64_t* count_out) {
  REQUIRE(c && count_out, RICADI_EINVAL, "NULL argument");
  *count_out = (int64_t)c->xcount;
  return RICADI_OK;
}

int ricadi_set_dims(ricadi_ctx* c, int nv) {
  REQUIRE(c && nv > 0, RICADI_EINVAL, "bad argument");
  c->cache.clear();
  for (auto& e : c->rec_ring) e->serial = -1;
  c->has_op = false;
  c->nv = nv;
  c->np = 0;
  c->n = nv;
  c->q = 0;
  c->zc = 0;
  c->wcols = 0;
  return RICADI_OK;
}

int ricadi_set_lowrank(ricadi_ctx* c, const double* U, const double* V, int q) {
  REQUIRE(c && c->has_op, RICADI_ESTATE, "set the operator first");
  REQUIRE(q >= 0 && q <= 64, RICADI_EINVAL, "low-rank width must be in [0, 64]");
  REQUIRE(q == 0 || (U && V), RICADI_EINVAL, "NULL low-rank factor");
  API_BEGIN
  c->q = q;
  ++c->lr_epoch;
  if (q > 0) {
    const size_t cnt = (size_t)c->nv * q;
    c->U.ensure(cnt);
    c->V.ensure(cnt);
    HIPCHK(hipMemcpyAsync(c->U.p, U, cnt * sizeof(double), hipMemcpyHostToDevice, c->st));
    HIPCHK(hipMemcpyAsync(c->V.p, V, cnt * sizeof(double), hipMemcpyHostToDevice, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
  }
  API_END
}

static int check_panel(ricadi_ctx* c, int m) {
  REQUIRE(c && c->has_op, RICADI_ESTATE, "set the operator first");
  (void)hipSetDevice(c->dev);   // host worker threads start on device 0
  REQUIRE(m >= 1 && m <= RICADI_MAX_M, RICADI_EINVAL, "panel width must be in [1, 128]");
  return RICADI_OK;
}

int ricadi_spmm_dev(ricadi_ctx* c, double alpha, double beta, const double* dX, int m, double* dY) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(dX && dY, RICADI_EINVAL, "NULL panel");
  API_BEGIN
  ShiftData* sd = get_shift(c, alpha, beta);
  ensure_work(c, m);
  op_apply(c, sd, dX, dY, m, true);
  API_END
}

int ricadi_spmm(ricadi_ctx* c, double alpha, double beta, const double* X, int m, double* Y) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(X && Y, RICADI_EINVAL, "NULL panel");
  API_BEGIN
  const size_t nm = (size_t)c->n * m;
  ShiftData* sd = get_shift(c, alpha, beta);
  ensure_work(c, m);
  HIPCHK(hipMemcpyAsync(c->pw1.p, X, nm * sizeof(double), hipMemcpyHostToDevice, c->st));
  op_apply(c, sd, c->pw1.p, c->pw2.p, m, true);
  HIPCHK(hipMemcpyAsync(Y, c->pw2.p, nm * sizeof(double), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  API_END
}

int ricadi_precond_apply(ricadi_ctx* c, double alpha, double beta, const double* R, int m, double* Z) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(R && Z, RICADI_EINVAL, "NULL panel");
  API_BEGIN
  const size_t nm = (size_t)c->n * m;
  ShiftData* sd = get_shift(c, alpha, beta);
  ensure_work(c, m);
  HIPCHK(hipMemcpyAsync(c->pw1.p, R, nm * sizeof(double), hipMemcpyHostToDevice, c->st));
  precond_apply(c, sd, c->pw1.p, c->pw2.p, m);
  HIPCHK(hipMemcpyAsync(Z, c->pw2.p, nm * sizeof(double), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  API_END
}

int ricadi_shift_solve_dev(ricadi_ctx* c, double alpha, double beta, const double* dR, int m,
                           double* dX, int* iters_out, double* relres_out) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(dR && dX, RICADI_EINVAL, "NULL panel");
  int status = RICADI_OK;
  try {
    ShiftData* sd = get_shift(c, alpha, beta);
    ensure_work(c, m);
    load_rhs(c, dR, m, c->bvec.p);
    GmresResult r = gmres_solve(c, sd, c->bvec.p, dX, m, true, relres_out);
    if (iters_out) *iters_out = r.iters;
    if (!r.converged) {
      ricadi::set_error("GMRES did not reach the tolerance");
      status = RICADI_ENOCONV;
    }
  } catch (const ricadi::HipError& e) {
    ricadi::set_error(e.msg);
    return RICADI_EHIP;
  } catch (const std::exception& e) {
    ricadi::set_error(e.what());
    return RICADI_EHIP;
  } catch (...) {
    ricadi::set_error("unknown C++ exception");
    return RICADI_EHIP;
  }
  return status;
}

int ricadi_shift_solve_batch_dev(ricadi_ctx* c, int ng, const double* alphas, const double* betas,
                                 const double* dR, int64_t r_stride, int m, double* dX,
                                 int* iters_out, double* relres_out) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(dR && dX && alphas && betas, RICADI_EINVAL, "NULL argument");
  REQUIRE(ng >= 1 && ng <= RICADI_MAX_GROUPS && (size_t)ng * m <= 2048, RICADI_EINVAL,
          "1 <= ng <= 16 and ng*m <= 2048 required");
  REQUIRE(r_stride == 0 || r_stride >= (int64_t)c->nv * m, RICADI_EINVAL, "bad r_stride");
  int status = RICADI_OK;
  try {
    std::vector<ShiftData*> sds(ng);
    get_shifts(c, alphas, betas, ng, sds.data());
    ensure_work(c, m, ng);
    const size_t nm = (size_t)c->n * m;
    const int nload = r_stride == 0 ? 1 : ng;
    for (int g = 0; g < nload; ++g) load_rhs(c, dR + (size_t)g * r_stride, m, c->bvec.p + (size_t)g * nm);
    std::vector<GmresResult> res(ng);
    solve_batch(c, sds.data(), ng, c->bvec.p, r_stride == 0 ? 0 : nm, dX, m, true, relres_out,
                res.data());
    for (int g = 0; g < ng; ++g) {
      if (iters_out) iters_out[g] = res[g].iters;
      if (!res[g].converged) {
        ricadi::set_error("GMRES did not reach the tolerance");
        status = RICADI_ENOCONV;
      }
    }
  } catch (const ricadi::HipError& e) {
    ricadi::set_error(e.msg);
    return RICADI_EHIP;
  } catch (const std::exception& e) {
    ricadi::set_error(e.what());
    return RICADI_EHIP;
  } catch (...) {
    ricadi::set_error("unknown C++ exception");
    return RICADI_EHIP;
  }
  return status;
}

int ricadi_shift_solve(ricadi_ctx* c, double alpha, double beta, const double* R, const double* Rp,
                       int m, double* X_out, int* iters_out, double* relres_out) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(R && X_out, RICADI_EINVAL, "NULL panel");
  int status = RICADI_OK;
  try {
    const size_t nm = (size_t)c->n * m, nvm = (size_t)c->nv * m;
    ShiftData* sd = get_shift(c, alpha, beta);
    ensure_work(c, m);
    HIPCHK(hipMemcpyAsync(c->bvec.p, R, nvm * sizeof(double), hipMemcpyHostToDevice, c->st));
    if (c->np > 0) {
      if (Rp)
        HIPCHK(hipMemcpyAsync(c->bvec.p + nvm, Rp, (nm - nvm) * sizeof(double), hipMemcpyHostToDevice, c->st));
      else
        HIPCHK(hipMemsetAsync(c->bvec.p + nvm, 0, (nm - nvm) * sizeof(double), c->st));
    }
    GmresResult r = gmres_solve(c, sd, c->bvec.p, c->xs.p, m, true, relres_out);
    HIPCHK(hipMemcpyAsync(X_out, c->xs.p, nm * sizeof(double), hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    if (iters_out) *iters_out = r.iters;
    if (!r.converged) {
      ricadi::set_error("GMRES did not reach the tolerance");
      status = RICADI_ENOCONV;
    }
  } catch (const ricadi::HipError& e) {
    ricadi::set_error(e.msg);
    return RICADI_EHIP;
  } catch (const std::exception& e) {
    ricadi::set_error(e.what());
    return RICADI_EHIP;
  } catch (...) {
    ricadi::set_error("unknown C++ exception");
    return RICADI_EHIP;
  }
  return status;
}

int ricadi_apply_e_dev(ricadi_ctx* c, double coef, const double* dV, int m, double* dW) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(dV && dW, RICADI_EINVAL, "NULL panel");
  API_BEGIN
  launch_spmm(c->st, c->nv, c->E.rp.p, c->E.ci.p, c->E.v.p, dV, m, nullptr, dW, m, dW, m, coef, 1.0,
              nullptr, m);
  API_END
}

int ricadi_lincomb_dev(ricadi_ctx* c, int nrows, int m, int nvec, const double* dBasis,
                       int64_t stride, const double* coef, double* dOut) {
  REQUIRE(c && dBasis && coef && dOut, RICADI_EINVAL, "NULL argument");
  REQUIRE(nrows > 0 && m >= 1 && m <= RICADI_MAX_M && nvec >= 1 && nvec <= 64, RICADI_EINVAL,
          "bad sizes");
  API_BEGIN
  std::vector<double> h((size_t)nvec * m);
  for (int i = 0; i < nvec; ++i)
    for (int j = 0; j < m; ++j) h[(size_t)i * m + j] = coef[i];
  c->scratch.ensure((size_t)nvec * m + 64);
  HIPCHK(hipMemcpyAsync(c->scratch.p, h.data(), sizeof(double) * nvec * m, hipMemcpyHostToDevice, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  launch_cols_update(c->st, nrows, m, nvec, dBasis, (size_t)stride, c->scratch.p, 1.0, nullptr, nullptr,
                     dOut);
  API_END
}

int ricadi_sweep_recombine_slots_dev(ricadi_ctx* c, int nslot, int G, const double* dU, int m,
                                     const double* coefz, const double* coefw, double* dZ, double* dW,
                                     double* n2_out, double* block_n2_out) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(dU && coefz && coefw && dZ && dW && n2_out, RICADI_EINVAL, "NULL argument");
  REQUIRE(G >= 1 && G <= 64 && nslot >= 1 && nslot <= 128 && G * m <= 2048, RICADI_EINVAL,
          "1 <= G <= 64, 1 <= nslot <= 128 and G*m <= 2048 required");
  API_BEGIN
  hipStream_t st = c->st;
  const int nv = c->nv;
  const size_t nvm = (size_t)nv * m;
  ensure_work(c, m, std::min(G, RICADI_MAX_GROUPS));
  c->sweep_t.ensure(nvm);
  c->sweep_coef.ensure((size_t)(G + 1) * nslot * m);
  c->scratch.ensure((size_t)G * m + 64);
  // coefficient rows replicated over the m columns: G columns of coefz, then coefw
  std::vector<double> coef((size_t)(G + 1) * nslot * m);
  for (int j = 0; j <= G; ++j)
    for (int i = 0; i < nslot; ++i) {
      const double v = j < G ? coefz[(size_t)i * G + j] : coefw[i];
      for (int cc = 0; cc < m; ++cc) coef[((size_t)j * nslot + i) * m + cc] = v;
    }
  HIPCHK(hipMemcpyAsync(c->sweep_coef.p, coef.data(), sizeof(double) * coef.size(),
                        hipMemcpyHostToDevice, st));
  // Z-block j = sum_i coefz[i][j] U_i  (columns j*m .. of dZ, leading dimension G*m)
  for (int j = 0; j < G; ++j) {
    launch_cols_update(st, nv, m, nslot, dU, nvm, c->sweep_coef.p + (size_t)j * nslot * m, 1.0, nullptr,
                       nullptr, c->sweep_t.p);
    launch_copy_cols(st, nv, m, c->sweep_t.p, m, 0, dZ, G * m, j * m, 1.0);
    col_norms2(c, c->sweep_t.p, nv, m, c->scratch.p + (size_t)j * m);
  }
  // W += E (sum_i coefw[i] U_i)
  launch_cols_update(st, nv, m, nslot, dU, nvm, c->sweep_coef.p + (size_t)G * nslot * m, 1.0, nullptr,
                     nullptr, c->sweep_t.p);
  launch_spmm(st, nv, c->E.rp.p, c->E.ci.p, c->E.v.p, c->sweep_t.p, m, nullptr, dW, m, dW, m, 1.0, 1.0,
              nullptr, m);
  std::vector<double> nr((size_t)G * m);
  HIPCHK(hipMemcpyAsync(nr.data(), c->scratch.p, sizeof(double) * G * m, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));   // also keeps `coef` alive until its upload has run
  double n2 = 0.0;
  for (double v : nr) n2 += v;
  *n2_out = n2;
  if (block_n2_out)
    for (int j = 0; j < G; ++j) {
      double b2 = 0.0;
      for (int cc = 0; cc < m; ++cc) b2 += nr[(size_t)j * m + cc];
      block_n2_out[j] = b2;
    }
  API_END
}

int ricadi_sweep_recombine_dev(ricadi_ctx* c, int G, const double* dU, int m, const double* rinv,
                               const double* cinv1, double* dZ, double* dW, double* n2_out) {
  return ricadi_sweep_recombine_slots_dev(c, G, G, dU, m, rinv, cinv1, dZ, dW, n2_out, nullptr);
}

int ricadi_gain_dev(ricadi_ctx* c, double coef, const double* dZ, int cz, int ldz, const double* dB,
                    int nb, double* dK) {
  REQUIRE(c && c->has_op, RICADI_ESTATE, "set the operator first");
  REQUIRE(dZ && dB && dK && cz > 0 && ldz >= cz && nb >= 1 && nb <= RICADI_MAX_M, RICADI_EINVAL,
          "bad argument");
  API_BEGIN
  gain_dev(c, c->E, dZ, cz, ldz, dB, nb, dK);
  if (coef != 1.0) launch_axpby(c->st, (size_t)c->nv * nb, coef, dK, 0.0, dK);
  HIPCHK(hipStreamSynchronize(c->st));
  API_END
}

int ricadi_panel_norms_dev(ricadi_ctx* c, const double* dW, int nrows, int m, double* gram_fro,
                           double* nrm2) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(dW && nrows > 0, RICADI_EINVAL, "bad panel");
  API_BEGIN
  DScalar::gram_norms(c, dW, nrows, m, gram_fro, nrm2);
  API_END
}

int ricadi_time_spmm_dev(ricadi_ctx* c, double alpha, double beta, const double* dX, int m,
                         double* dY, int reps, double* ms_per_launch) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(dX && dY && reps > 0 && ms_per_launch, RICADI_EINVAL, "bad argument");
  API_BEGIN
  ShiftData* sd = get_shift(c, alpha, beta);
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  // plain assembled-CSR saddle SpMM only (no low-rank term): the roofline kernel
  HIPCHK(hipEventRecord(e0, c->st));
  const Batch bt = make_batch(c, sd, m);
  for (int i = 0; i < reps; ++i) saddle_spmm(c, bt, dX, bt.gs, nullptr, dY, bt.gs, nullptr, 0, 1.0, 0.0);
  HIPCHK(hipEventRecord(e1, c->st));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_per_launch = (double)ms / reps;
  API_END
}

int ricadi_time_spmm_batch_dev(ricadi_ctx* c, int ng, const double* alphas, const double* betas,
                               const double* dX, int m, double* dY, int reps, double* ms_per_launch) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(dX && dY && alphas && betas && reps > 0 && ms_per_launch, RICADI_EINVAL, "bad argument");
  REQUIRE(ng >= 1 && ng <= RICADI_MAX_GROUPS, RICADI_EINVAL, "1 <= ng <= 16 required");
  API_BEGIN
  std::vector<ShiftData*> sds(ng);
  get_shifts(c, alphas, betas, ng, sds.data());
  const Batch bt = make_batch(c, sds.data(), ng, m);
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  // the saddle SpMM exactly as the batched GMRES launches it (no low-rank term; on the FP32-stored vector
  // when the iteration does so)
  DArr<float> x32;
  if (iteration_reads_x32(c, m, ng)) {
    x32.alloc(bt.gs * ng);
    for (int g = 0; g < ng; ++g)
      launch_to_f32(c->st, c->n, m, dX + (size_t)g * bt.gs, m, x32.p + (size_t)g * bt.gs, m);
  }
  // ... and into an FP32 panel when the iteration's Arnoldi passes read one (dY is then left alone)
  DArr<float> y32;
  const bool b16t = getenv("RICADI_BASIS64") == nullptr && getenv("RICADI_BASIS32") == nullptr && c->n <= (1 << 21);
  if (x32.p && iteration_w32(c, m, ng, b16t, update_hess_fused_ok(m, b16t),
                             update_dots_keeps_w(m, b16t, c->opts.gmres_restart), c->opts.gmres_restart))
    y32.alloc(bt.gs * ng);
  c->w32_last = y32.p ? 1 : 0;
  saddle_spmm(c, bt, dX, bt.gs, nullptr, dY, bt.gs, nullptr, 0, 1.0, 0.0, LowRankArgs(), x32.p, y32.p);
  HIPCHK(hipEventRecord(e0, c->st));
  for (int i = 0; i < reps; ++i)
    saddle_spmm(c, bt, dX, bt.gs, nullptr, dY, bt.gs, nullptr, 0, 1.0, 0.0, LowRankArgs(), x32.p, y32.p);
  HIPCHK(hipEventRecord(e1, c->st));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_per_launch = (double)ms / reps;
  API_END
}

// One launch (or launch pair: the dot kernels come with their partial-sum reduction) of a
// hot-path kernel class exactly as the batched GMRES issues it, timed with HIP events on
// the context stream.  Operands are the solver's own workspace buffers, filled with finite
// values; results are discarded.
int ricadi_time_kernel_dev(ricadi_ctx* c, int which, int ng, const double* alphas, const double* betas,
                           int m, int nvec, int reps, double* ms_per_launch) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(alphas && betas && reps > 0 && ms_per_launch, RICADI_EINVAL, "bad argument");
  REQUIRE(ng >= 1 && ng <= RICADI_MAX_GROUPS && (size_t)ng * m <= 2048, RICADI_EINVAL,
          "1 <= ng <= 16 and ng*m <= 2048 required");
  REQUIRE(nvec >= 1 && nvec <= c->opts.gmres_restart, RICADI_EINVAL, "1 <= nvec <= gmres_restart required");
  API_BEGIN
  hipStream_t st = c->st;
  std::vector<ShiftData*> sds(ng);
  get_shifts(c, alphas, betas, ng, sds.data());
  ensure_work(c, m, ng, 0);
  Batch bt = make_batch(c, sds.data(), ng, m);
  const int n = c->n, restart = c->opts.gmres_restart;
  const size_t nm = bt.gs, vs = nm * ng;
  const size_t gsh = (size_t)(restart + 2) * m;
  const size_t gspart = (size_t)dots_num_blocks(n) * (restart + 2) * m;
  // finite fill: byte 0x3C -> 1.5e-18 (FP64), 1.06 (FP16), 0.0115 (FP32)
  HIPCHK(hipMemsetAsync(c->wv.p, 0x3C, sizeof(double) * vs, st));
  HIPCHK(hipMemsetAsync(c->zv.p, 0x3C, sizeof(double) * vs, st));
  if (c->zbasisf.p) HIPCHK(hipMemsetAsync(c->zbasisf.p, 0x3C, sizeof(float) * vs, st));
  HIPCHK(hipMemsetAsync(c->r2.p, 0x3C, sizeof(double) * vs, st));
  HIPCHK(hipMemsetAsync(c->h1.p, 0x3C, sizeof(double) * gsh * ng, st));
  HIPCHK(hipMemsetAsync(c->h2.p, 0x3C, sizeof(double) * gsh * ng, st));
  HIPCHK(hipMemsetAsync(c->scale.p, 0x3C, sizeof(double) * (size_t)ng * m, st));
  HIPCHK(hipMemsetAsync(c->resid.p, 0x3C, sizeof(double) * 2 * c->wcols, st));
  HIPCHK(hipMemsetAsync(c->bnorm2.p, 0x3C, sizeof(double) * (size_t)ng * m, st));
  HIPCHK(hipMemsetAsync(c->g.p, 0x3C, sizeof(double) * (size_t)ng * m * (restart + 1), st));
  HIPCHK(hipMemsetAsync(c->cs.p, 0x3C, sizeof(double) * (size_t)ng * m * restart, st));
  HIPCHK(hipMemsetAsync(c->sn.p, 0x3C, sizeof(double) * (size_t)ng * m * restart, st));
  if (c->kc > 0) {
    HIPCHK(hipMemsetAsync(c->rc.p, 0x3C, sizeof(double) * bt.gsc * ng, st));
    HIPCHK(hipMemsetAsync(c->ec.p, 0x3C, sizeof(double) * bt.gsc * ng, st));
  }
  if (c->np > 0) HIPCHK(hipMemsetAsync(c->tp.p, 0x3C, sizeof(double) * bt.gsp * ng, st));
  const bool b16 = c->basis16, b32 = c->basis32 && !b16;
  const size_t basis_bytes = (size_t)(nvec + 1) * vs * (b16 ? 2 : b32 ? 4 : 8);
  if (c->basis32) {
    HIPCHK(hipMemsetAsync(c->basisf.p, 0x3C, basis_bytes, st));
    HIPCHK(hipMemsetAsync(c->vcur.p, 0x3C, sizeof(double) * vs, st));
  } else {
    HIPCHK(hipMemsetAsync(c->basis.p, 0x3C, basis_bytes, st));
  }
  _Float16* Vh = reinterpret_cast<_Float16*>(c->basisf.p);
  float* Vf = c->basisf.p;
  double* V = c->basis.p;
  const GroupTab& gt = bt.tab;
  const GroupPtrs ones = same_ptr(c->ones.p);
  // the operator's output and the Arnoldi passes on the FP32 panel where the iteration uses it
  const bool tw32 = b16 && c->zbasisf.p &&
                    iteration_w32(c, m, ng, b16, update_hess_fused_ok(m, b16), update_dots_keeps_w(m, b16, restart), restart);
  c->w32_last = tw32 ? 1 : 0;
  auto launch = [&]() {
    switch (which) {
      case 0:
        saddle_spmm(c, bt, c->zv.p, nm, nullptr, c->wv.p, nm, nullptr, 0, 1.0, 0.0, LowRankArgs(),
                    iteration_reads_x32(c, m, ng) && c->zbasisf.p ? c->zbasisf.p : nullptr, tw32 ? c->wv32.p : nullptr);
        break;
      case 1:
        if (c->precond32)
          launch_block_apply_b(st, gt, c->bs, c->nbv, c->bv_ptr.p, c->bv_rows.p, bt.bvinvf, c->r2.p, m, nm,
                               c->zv.p, m, nm, m, 0);
        else
          launch_block_apply_b(st, gt, c->bs, c->nbv, c->bv_ptr.p, c->bv_rows.p, bt.bvinv, c->r2.p, m, nm,
                               c->zv.p, m, nm, m, 0);
        break;
      case 2:
        if (c->nbp <= 0) throw HipError{"no pressure block"};
        if (c->precond32)
          launch_block_apply_b(st, gt, c->bs, c->nbp, c->bp_ptr.p, c->bp_rows.p, bt.bpinvf, c->tp.p, m,
                               bt.gsp, c->zv.p + (size_t)c->nv * m, m, nm, m, 0);
        else
          launch_block_apply_b(st, gt, c->bs, c->nbp, c->bp_ptr.p, c->bp_rows.p, bt.bpinv, c->tp.p, m,
                               bt.gsp, c->zv.p + (size_t)c->nv * m, m, nm, m, 0);
        break;
      case 3:
        if (c->kc <= 0) throw HipError{"no coarse level"};
        {
          // the dense inverse lives on the last level
          ricadi_ctx* lc = c;
          Batch lb = bt;
          while (lc->child) {
            Batch t = *lb.sub;
            t.tab = gt;
            lb = t;
            lc = lc->child.get();
          }
          if (c->precond32)
            launch_dense_apply_b(st, gt, lc->kc, m, lb.einvf, (lc->kc + 3) & ~3, lc->rc.p, lc->ec.p, c->coarse_mfma32);
          else
            launch_dense_apply_b(st, gt, lc->kc, m, lb.einv, lc->rc.p, lc->ec.p);
        }
        break;
      case 4:
        if (!c->syb_ok) throw HipError{"no tiled S*Y"};
        if (ms_pays(c, gt.ng, c->snnz) && spmm_blocked_ms_ok(m, c->syb_max_cols, (size_t)c->kc))
          launch_spmm_blocked_ms(st, gt, bt.alpha, bt.beta, c->sb_nblk, c->sb_rows2.p, c->syb_rp2.p,
                                 c->syb_cols2.p, c->syb_lidx_ms.p, c->sybAJ.p, c->sybE.p, c->ec.p,
                                 m, bt.gsc, c->r2.p, m, nm, c->wv.p, m, nm, -1.0, 1.0, m, c->syb_max_cols);
        else
        launch_spmm_blocked_b(st, gt, c->sb_nblk, c->sb_rows2.p, c->syb_rp2.p, c->syb_cols2.p,
                              c->syb_lidx.p, bt.syvalb, c->ec.p, m, bt.gsc, c->r2.p, m, nm, c->wv.p, m, nm,
                              -1.0, 1.0, m, c->syb_max_cols);
        break;
      case 5:
        if (tw32) launch_cols_dots16_w32(st, gt, n, nvec, Vh, vs, nm, c->wv32.p, nm, c->partial.p, gspart, c->h1.p, gsh);
        else if (b16) launch_cols_dots_b(st, gt, n, m, nvec, Vh, vs, nm, c->wv.p, nm, 0, c->partial.p, gspart, c->h1.p, gsh);
        else if (b32) launch_cols_dots_b(st, gt, n, m, nvec, Vf, vs, nm, c->wv.p, nm, 0, c->partial.p, gspart, c->h1.p, gsh);
        else launch_cols_dots_b(st, gt, n, m, nvec, V, vs, nm, c->wv.p, nm, 0, c->partial.p, gspart, c->h1.p, gsh);
        break;
      case 6:
        set_update_dots_nostore(update_dots_keeps_w(m, b16, restart));   // as the iteration launches it
        if (tw32) launch_cols_update_dots16_w32(st, gt, n, nvec, Vh, vs, nm, c->h1.p, gsh, c->wv32.p, nm, c->partial.p, gspart, c->h2.p, gsh);
        else if (b16) launch_cols_update_dots_b(st, gt, n, m, nvec, Vh, vs, nm, c->h1.p, gsh, c->wv.p, nm, c->partial.p, gspart, c->h2.p, gsh);
        else if (b32) launch_cols_update_dots_b(st, gt, n, m, nvec, Vf, vs, nm, c->h1.p, gsh, c->wv.p, nm, c->partial.p, gspart, c->h2.p, gsh);
        else launch_cols_update_dots_b(st, gt, n, m, nvec, V, vs, nm, c->h1.p, gsh, c->wv.p, nm, c->partial.p, gspart, c->h2.p, gsh);
        set_update_dots_nostore(false);
        break;
      case 7:
        if (b16 && update_hess_fused_ok(m, b16))     // as the iteration launches it: with the Hessenberg update
          launch_cols_update16_hess_b(st, gt, n, nvec, Vh, vs, nm, c->h1.p, c->h2.p, gsh, update_dots_keeps_w(m, b16, restart) ? 1 : 0,
                                      c->wv.p, nm, precond_reads_h16(c, m) ? nullptr : c->vcur.p, nm, Vh + (size_t)nvec * vs, nm,
                                      nvec - 1, restart, c->H.p, c->cs.p, c->sn.p, c->g.p, c->resid.p, c->resid.p + c->wcols,
                                      c->bnorm2.p, c->opts.gmres_tol, nullptr, tw32 ? c->wv32.p : nullptr);
        else if (b16) launch_cols_update_b(st, gt, n, m, nvec, Vh, vs, nm, c->h2.p, gsh, -1.0, c->wv.p, nm, c->scale.p, precond_reads_h16(c, m) ? nullptr : c->vcur.p, nm, Vh + (size_t)nvec * vs, nm);
        else if (b32) launch_cols_update_b(st, gt, n, m, nvec, Vf, vs, nm, c->h2.p, gsh, -1.0, c->wv.p, nm, c->scale.p, c->vcur.p, nm, Vf + (size_t)nvec * vs, nm);
        else launch_cols_update_b(st, gt, n, m, nvec, V, vs, nm, c->h2.p, gsh, -1.0, c->wv.p, nm, c->scale.p, V + (size_t)nvec * vs, nm);
        break;
      case 8:
        if (b16 && precond_reads_h16(c, m))
          precond_apply(c, bt, c->wv.p, nm, c->zv.p, c->zbasisf.p, nm, iteration_reads_x32(c, m, ng), Vh);
        else
          precond_apply(c, bt, c->wv.p, nm, c->zv.p, c->zbasisf.p, nm, iteration_reads_x32(c, m, ng));
        break;
      case 9:
        if (c->kc <= 0) throw HipError{"no coarse level"};
        launch_spmm_b(st, gt, c->kc, c->agg_ptr.p, c->agg_rows.p, ones, c->wv.p, m, nm, nullptr, c->rc.p, m,
                      bt.gsc, nullptr, 0, 0, 1.0, 0.0, m);
        break;
      case 10: case 11: case 12: case 13: case 14: case 15: case 16: {
        // ONE stage of the preconditioner application, through the launcher precond_apply itself uses
        Restore<int> keep(c->pc_stage);
        c->pc_stage = which - 10;
        if (b16 && precond_reads_h16(c, m))
          precond_apply(c, bt, c->wv.p, nm, c->zv.p, c->zbasisf.p, nm, iteration_reads_x32(c, m, ng), Vh);
        else
          precond_apply(c, bt, c->wv.p, nm, c->zv.p, c->zbasisf.p, nm, iteration_reads_x32(c, m, ng));
        break;
      }
      default:
        throw HipError{"unknown kernel class"};
    }
  };
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  launch();   // warm-up (code object load, caches)
  HIPCHK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) launch();
  HIPCHK(hipEventRecord(e1, st));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_per_launch = (double)ms / reps;
  API_END
}

int ricadi_qr(ricadi_ctx* c, const double* Z, int cz, double* Q_out, double* R_out) {
  REQUIRE(c && c->nv > 0, RICADI_ESTATE, "set the operator (or the dimensions) first");
  REQUIRE(Z && R_out && cz > 0 && cz <= c->nv, RICADI_EINVAL, "bad argument");
  API_BEGIN
  (void)hipSetDevice(c->dev);
  const int nv = c->nv;
  DArr<double> dZ, dQ, dR;
  dZ.alloc((size_t)nv * cz);
  dQ.alloc((size_t)nv * cz);
  dR.alloc((size_t)cz * cz);
  HIPCHK(hipMemcpyAsync(dZ.p, Z, sizeof(double) * nv * cz, hipMemcpyHostToDevice, c->st));
  block_qr_dev(c, dZ.p, cz, nv, cz, dQ.p, dR.p);
  HIPCHK(hipMemcpyAsync(R_out, dR.p, sizeof(double) * cz * cz, hipMemcpyDeviceToHost, c->st));
  if (Q_out) HIPCHK(hipMemcpyAsync(Q_out, dQ.p, sizeof(double) * nv * cz, hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  API_END
}

int ricadi_setup_info(ricadi_ctx* c, int* out, int nout) {
  REQUIRE(c && out && nout >= 8, RICADI_EINVAL, "bad argument");
  out[0] = c->nv;
  out[1] = c->np;
  out[2] = c->nbv;
  out[3] = c->nbp;
  out[4] = c->bs;
  out[5] = c->kc;
  out[6] = c->sb_nblk;
  out[7] = c->sb_max_cols;
  for (int i = 8; i < nout; ++i) out[i] = 0;
  // [8]: levels in use; [9]: size of the dense inverse on the last level
  int lv = c->kc > 0 ? 2 : 1;
  const ricadi_ctx* lc = c;
  for (; lc->child; lc = lc->child.get()) ++lv;
  if (nout > 8) out[8] = lv;
  if (nout > 9) out[9] = lc->kc;
  // [10]: 1 if the iteration reads the current vector from the FP16 basis (no FP64 copy written), 16-column panels
  if (nout > 10) out[10] = (c->has_op && (getenv("RICADI_BASIS64") == nullptr) && (getenv("RICADI_BASIS32") == nullptr) &&
                            c->n <= (1 << 21) && precond_reads_h16_static(c)) ? 1 : 0;
  // [11], [12]: padded widths of the dense rectangles of the last / first velocity sweep (0: sweep not in that form);
  // [13]: pressure dofs per Schur block list entry count (np), [14]: nnz(J), [15]: nnz of the pressure rows of S*Y
  if (nout > 11) out[11] = c->gt_ok ? c->gt_ks : 0;
  if (nout > 12) out[12] = (c->ady_ok && c->kc > 0) ? c->ady_ks : 0;
  if (nout > 13) out[13] = c->np;
  if (nout > 14) out[14] = (int)c->J.ci.n;
  if (nout > 15) out[15] = c->kc > 0 && c->np > 0 ? (int)(c->synnz) : 0;
  // [16]: entries of the restriction (rows of P^T with smoothed aggregation; else one per dof)
  if (nout > 16) out[16] = c->kc > 0 ? (c->sa ? (int)c->pt_ci.n : c->n) : 0;
  // [17]: route of the last batch of dense coarse inverses on the last level (0 block Gauss-Jordan, 1 rocSOLVER with
  // partial pivoting; -1 none yet); [18]: kernel of the last saddle SpMM launch (0 CSR, 1 LDS-tiled per
  // group, 2 LDS-tiled multi-shift, +4: FP32 x input; -1 none yet)
  if (nout > 17) out[17] = lc->coarse_route;
  if (nout > 18) out[18] = c->k1_variant;
  // [19]: the last preconditioner application kept the velocity part between its sweeps as an FP32 panel (1) or as
  // an FP64 panel (0); -1 none yet
  if (nout > 19) out[19] = c->mid32_last;
  // [20]: the operator launch of the last iteration / timing call wrote w as an FP32 panel (1) or FP64 (0); -1 none yet
  if (nout > 20) out[20] = c->w32_last;
  return RICADI_OK;
}

int ricadi_dense_inverse_batch(ricadi_ctx* c, int k, int nb, double* A, int* route_out) {
  REQUIRE(c && A && k >= 1 && nb >= 1 && nb <= 4 * RICADI_MAX_GROUPS, RICADI_EINVAL, "bad argument");
  API_BEGIN
  hipStream_t st = c->st;
  const size_t kk = (size_t)k * k;
  DArr<double> dA, dA0;
  dA.alloc(kk * nb);
  dA0.alloc(kk * nb);
  HIPCHK(hipMemcpyAsync(dA.p, A, sizeof(double) * kk * nb, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(dA0.p, dA.p, sizeof(double) * kk * nb, hipMemcpyDeviceToDevice, st));
  std::vector<double*> hp(nb);
  for (int i = 0; i < nb; ++i) hp[i] = dA.p + kk * i;
  std::vector<int> info(nb, 0);
  const int route = invert_dense_batch(c, hp, k, info, [&] {
    HIPCHK(hipMemcpyAsync(dA.p, dA0.p, sizeof(double) * kk * nb, hipMemcpyDeviceToDevice, st));
  });
  if (route_out) *route_out = route;
  for (int i = 0; i < nb; ++i)
    if (info[i] != 0) throw HipError{"matrix " + std::to_string(i) + " singular (getrf/getri info " + std::to_string(info[i]) + ")"};
  HIPCHK(hipMemcpyAsync(A, dA.p, sizeof(double) * kk * nb, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  API_END
}

int ricadi_time_qr_dev(ricadi_ctx* c, const double* dZ, int cz, int reps, double* ms_per_call) {
  REQUIRE(c && c->nv > 0 && dZ && cz > 0 && cz <= c->nv && reps > 0 && ms_per_call, RICADI_EINVAL,
          "bad argument");
  API_BEGIN
  (void)hipSetDevice(c->dev);
  DArr<double> Q, R;
  Q.alloc((size_t)c->nv * cz);
  R.alloc((size_t)cz * cz);
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, c->st));
  for (int i = 0; i < reps; ++i) block_qr_dev(c, dZ, cz, c->nv, cz, Q.p, R.p);
  HIPCHK(hipEventRecord(e1, c->st));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_per_call = (double)ms / reps;
  API_END
}

int ricadi_time_gram_dev(ricadi_ctx* c, const double* dZ, int cz, double* dG, int reps,
                         double* ms_per_launch) {
  REQUIRE(c && c->nv > 0 && dZ && dG && cz > 0 && reps > 0 && ms_per_launch, RICADI_EINVAL,
          "bad argument");
  API_BEGIN
  (void)hipSetDevice(c->dev);
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipMemsetAsync(dG, 0, sizeof(double) * cz * cz, c->st));
  HIPCHK(hipEventRecord(e0, c->st));
  for (int i = 0; i < reps; ++i) launch_gemm_tn(c->st, c->nv, cz, cz, dZ, cz, dZ, cz, dG, cz);
  HIPCHK(hipEventRecord(e1, c->st));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_per_launch = (double)ms / reps;
  API_END
}

int ricadi_lyap_adi(ricadi_ctx* c, const double* shifts, int ns, const double* W, int m,
                    const ricadi_adi_params* prm, double* Z_out, int* c_out, double* stats_out) {
  if (int rc = check_panel(c, m)) return rc;
  REQUIRE(shifts && ns > 0 && W && prm, RICADI_EINVAL, "bad argument");
  for (int i = 0; i < ns; ++i) REQUIRE(shifts[i] < 0.0, RICADI_EINVAL, "ADI shifts must be negative");
  REQUIRE(prm->adi_max_steps > 0, RICADI_EINVAL, "adi_max_steps must be positive");
  API_BEGIN
  ensure_work(c, m);
  const long esc0 = c->escalations;
  factor_reserve(c, prm->adi_max_steps * m);
  DArr<double> dW;
  dW.alloc((size_t)c->nv * m);
  HIPCHK(hipMemcpyAsync(dW.p, W, sizeof(double) * c->nv * m, hipMemcpyHostToDevice, c->st));
  if (c->timing) c->t_setup = c->t_solve = c->t_recomb = c->t_compress = c->t_proj = c->t_cyc = c->t_iter = c->t_guess = 0;
  Tick tka;
  AdiStats s = lyap_adi_dev(c, shifts, ns, dW.p, m, *prm);
  if (c->timing) {
    (void)hipStreamSynchronize(c->st);
    fprintf(stderr, "[ricadi timing] lyap_adi: total %.1f ms = setup %.1f + projection %.1f + solves %.1f (Arnoldi iterations %.1f, "
            "restart-cycle bookkeeping %.1f, recycled guesses %.1f) + recombination %.1f + recompression %.1f (+ rest)\n",
            1e3 * tka.lap(), 1e3 * c->t_setup, 1e3 * c->t_proj, 1e3 * c->t_solve, 1e3 * c->t_iter, 1e3 * c->t_cyc,
            1e3 * c->t_guess, 1e3 * c->t_recomb, 1e3 * c->t_compress);
  }
  if (c_out) *c_out = c->zc;
  if (Z_out && c->zc > 0) {
    HIPCHK(hipMemcpy2DAsync(Z_out, sizeof(double) * c->zc, c->Z.p, sizeof(double) * c->zld,
                            sizeof(double) * c->zc, c->nv, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
  }
  if (stats_out) {
    stats_out[0] = s.steps;
    stats_out[1] = s.rel;
    stats_out[2] = (double)s.gmres_iters;
    stats_out[3] = (double)s.shift_solves;
    stats_out[4] = s.res_fro;
    stats_out[5] = (double)s.nonconverged;
    stats_out[6] = s.worst_relres;
    stats_out[7] = (double)(c->escalations - esc0);
  }
  API_END
}

}  // extern "C"

extern "C" {

int ricadi_ric_newtonadi(ricadi_ctx* c, const double* shifts, int ns, const double* B, int nb,
                         const double* W, int mw, const double* Z0, int c0, const double* oldB,
                         const ricadi_adi_params* prm, double* Z_out, int zcap, int* c_out,
                         double* stats_out) {
  REQUIRE(c && c->has_op, RICADI_ESTATE, "set the operator first");
  REQUIRE(shifts && ns > 0 && B && W && prm, RICADI_EINVAL, "bad argument");
  REQUIRE(nb >= 1 && nb <= 64 && mw >= 1 && mw + nb <= RICADI_MAX_M, RICADI_EINVAL, "bad widths");
  REQUIRE(c0 == 0 || Z0, RICADI_EINVAL, "Z0 is NULL");
  for (int i = 0; i < ns; ++i) REQUIRE(shifts[i] < 0.0, RICADI_EINVAL, "ADI shifts must be negative");
  API_BEGIN
  hipStream_t st = c->st;
  const int nv = c->nv;
  TArr<double> dB(c->pool, (size_t)nv * nb), dWm(c->pool, (size_t)nv * mw), dOld(c->pool), dZ0(c->pool);
  HIPCHK(hipMemcpyAsync(dB.p, B, sizeof(double) * nv * nb, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(dWm.p, W, sizeof(double) * nv * mw, hipMemcpyHostToDevice, st));
  if (oldB) {
    dOld.alloc((size_t)nv * nb);
    HIPCHK(hipMemcpyAsync(dOld.p, oldB, sizeof(double) * nv * nb, hipMemcpyHostToDevice, st));
  }
  if (c0 > 0) {
    dZ0.alloc((size_t)nv * c0);
    HIPCHK(hipMemcpyAsync(dZ0.p, Z0, sizeof(double) * nv * c0, hipMemcpyHostToDevice, st));
  }
  ric_newtonadi_run(c, shifts, ns, dB.p, nb, dWm.p, mw, dZ0.p, c0, oldB ? dOld.p : nullptr, prm, stats_out);
  if (c_out) *c_out = c->zc;
  if (Z_out && c->zc > 0) {
    if (c->zc > zcap) throw ricadi::HipError{"Z_out capacity too small"};
    HIPCHK(hipMemcpy2DAsync(Z_out, sizeof(double) * c->zc, c->Z.p, sizeof(double) * c->zld,
                            sizeof(double) * c->zc, nv, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  API_END
}

// The same with every panel ALREADY ON THE DEVICE (no PCIe traffic inside the call): dB, dW, dZ0, dOldB are device
// pointers (row-major, leading dimension = width; dZ0 / dOldB may be NULL with c0 = 0).  The new iterate stays in
// the context's factor: ricadi_factor_cols, ricadi_factor_get (host) / ricadi_factor_get_dev (device).
int ricadi_ric_newtonadi_dev(ricadi_ctx* c, const double* shifts, int ns, const double* dB, int nb,
                             const double* dW, int mw, const double* dZ0, int c0, const double* dOldB,
                             const ricadi_adi_params* prm, int* c_out, double* stats_out) {
  REQUIRE(c && c->has_op, RICADI_ESTATE, "set the operator first");
  REQUIRE(shifts && ns > 0 && dB && dW && prm, RICADI_EINVAL, "bad argument");
  REQUIRE(nb >= 1 && nb <= 64 && mw >= 1 && mw + nb <= RICADI_MAX_M, RICADI_EINVAL, "bad widths");
  REQUIRE(c0 == 0 || dZ0, RICADI_EINVAL, "Z0 is NULL");
  for (int i = 0; i < ns; ++i) REQUIRE(shifts[i] < 0.0, RICADI_EINVAL, "ADI shifts must be negative");
  API_BEGIN
  ric_newtonadi_run(c, shifts, ns, dB, nb, dW, mw, dZ0, c0, dOldB, prm, stats_out);
  if (c_out) *c_out = c->zc;
  HIPCHK(hipStreamSynchronize(c->st));
  API_END
}

// Copy of the device-resident factor into a DEVICE buffer (nv x cz row-major, ld cz; cz = ricadi_factor_cols)
int ricadi_factor_get_dev(ricadi_ctx* c, double* dZ_out, int cz) {
  REQUIRE(c && dZ_out, RICADI_EINVAL, "NULL argument");
  REQUIRE(cz == c->zc && cz > 0, RICADI_EINVAL, "column count differs from the resident factor");
  API_BEGIN
  launch_copy_cols(c->st, c->nv, cz, c->Z.p, c->zld, 0, dZ_out, cz, 0, 1.0);
  HIPCHK(hipStreamSynchronize(c->st));
  API_END
}

int ricadi_compress(ricadi_ctx* c, const double* Z, int cz, double thresh, int kmax, double* Zc_out,
                    int* k_out, double* sv_out) {
  REQUIRE(c && c->nv > 0, RICADI_ESTATE, "set the operator (or the dimensions) first");
  REQUIRE(Zc_out && k_out, RICADI_EINVAL, "NULL output");
  API_BEGIN
  const double* dZ;
  int ld;
  DArr<double> tmp, out;
  if (Z) {
    REQUIRE(cz > 0, RICADI_EINVAL, "bad column count");
    tmp.alloc((size_t)c->nv * cz);
    HIPCHK(hipMemcpyAsync(tmp.p, Z, sizeof(double) * c->nv * cz, hipMemcpyHostToDevice, c->st));
    dZ = tmp.p;
    ld = cz;
  } else {
    REQUIRE(c->zc > 0, RICADI_ESTATE, "no device-resident factor");
    dZ = c->Z.p;
    cz = c->zc;
    ld = c->zld;
  }
  out.alloc((size_t)c->nv * cz);
  std::vector<double> sv;
  // the reference's route -- thin QR, then SVD of R ("QR ... SVD", optcont_main.py:133-134) -- up to 1024 columns
  // (the factors the Newton iteration returns are recompressed to a few hundred); raw factors beyond that take the
  // Gram route (singular values resolved to sqrt(eps) sigma_1 instead of eps sigma_1): an O(n c^2) block QR with
  // re-orthogonalisation of thousands of columns costs seconds
  const bool qr_route = c->opts.compress_qr != 0 && cz <= 1024;
  int k = compress_dev(c, dZ, cz, ld, thresh, kmax, false, out.p, &sv, qr_route);
  *k_out = k;
  if (k > 0) {
    HIPCHK(hipMemcpyAsync(Zc_out, out.p, sizeof(double) * c->nv * k, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
  }
  if (sv_out) std::memcpy(sv_out, sv.data(), sizeof(double) * std::min<size_t>(sv.size(), (size_t)std::min(cz, c->nv)));
  API_END
}

int ricadi_recompress(ricadi_ctx* c, const double* Z, int cz, double rel, double* Zc_out, int* k_out) {
  REQUIRE(c && c->nv > 0, RICADI_ESTATE, "set the operator (or the dimensions) first");
  REQUIRE(Z && Zc_out && k_out && cz > 0, RICADI_EINVAL, "NULL argument or bad column count");
  API_BEGIN
  DArr<double> tmp, out;
  tmp.alloc((size_t)c->nv * cz);
  out.alloc((size_t)c->nv * cz);
  HIPCHK(hipMemcpyAsync(tmp.p, Z, sizeof(double) * c->nv * cz, hipMemcpyHostToDevice, c->st));
  const int k = recompress_exec(c, main_exec(c), tmp.p, cz, cz, rel > 0.0 ? rel : kInternalRelThresh, out.p);
  *k_out = k;
  if (k > 0) {
    HIPCHK(hipMemcpyAsync(Zc_out, out.p, sizeof(double) * c->nv * k, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
  }
  API_END
}

int ricadi_gain(ricadi_ctx* c, const int32_t* mt_rp, const int32_t* mt_ci, const double* mt_v,
                const double* Z, int cz, const double* B, int nb, double* K_out) {
  REQUIRE(c && c->nv > 0, RICADI_ESTATE, "set the operator (or the dimensions) first");
  REQUIRE(mt_rp || c->has_op, RICADI_ESTATE, "no cal E in the context: pass mt_* explicitly");
  REQUIRE(B && K_out && nb >= 1 && nb <= RICADI_MAX_M, RICADI_EINVAL, "bad argument");
  API_BEGIN
  const int nv = c->nv;
  DArr<double> dZ, dB, dK;
  const double* z;
  int ld;
  if (Z) {
    REQUIRE(cz > 0, RICADI_EINVAL, "bad column count");
    dZ.alloc((size_t)nv * cz);
    HIPCHK(hipMemcpyAsync(dZ.p, Z, sizeof(double) * nv * cz, hipMemcpyHostToDevice, c->st));
    z = dZ.p;
    ld = cz;
  } else {
    REQUIRE(c->zc > 0, RICADI_ESTATE, "no device-resident factor");
    z = c->Z.p;
    cz = c->zc;
    ld = c->zld;
  }
  dB.alloc((size_t)nv * nb);
  dK.alloc((size_t)nv * nb);
  HIPCHK(hipMemcpyAsync(dB.p, B, sizeof(double) * nv * nb, hipMemcpyHostToDevice, c->st));
  if (mt_rp) {
    HostCsr Mt = make_csr(nv, nv, mt_rp, mt_ci, mt_v);
    DevCsr dMt;
    dMt.upload(Mt, c->st);
    gain_dev(c, dMt, z, cz, ld, dB.p, nb, dK.p);
  } else {
    gain_dev(c, c->E, z, cz, ld, dB.p, nb, dK.p);
  }
  HIPCHK(hipMemcpyAsync(K_out, dK.p, sizeof(double) * nv * nb, hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  API_END
}

int ricadi_lyap_res_norm(ricadi_ctx* c, const double* Z, int cz, const double* W, int m,
                         double* res2_out) {
  REQUIRE(c && c->has_op, RICADI_ESTATE, "set the operator first");
  REQUIRE(Z && W && res2_out && cz > 0 && m > 0, RICADI_EINVAL, "bad argument");
  API_BEGIN
  hipStream_t st = c->st;
  const int nv = c->nv, wtot = 2 * cz + m;
  DArr<double> dZ, S, chunk, G;
  dZ.alloc((size_t)nv * cz);
  S.alloc((size_t)nv * wtot);       // [cal A_eff Z, cal E Z, W], ld = wtot
  HIPCHK(hipMemcpyAsync(dZ.p, Z, sizeof(double) * nv * cz, hipMemcpyHostToDevice, st));
  {
    DArr<double> dWh;
    dWh.alloc((size_t)nv * m);
    HIPCHK(hipMemcpyAsync(dWh.p, W, sizeof(double) * nv * m, hipMemcpyHostToDevice, st));
    launch_copy_cols(st, nv, m, dWh.p, m, 0, S.p, wtot, 2 * cz, 1.0);
    HIPCHK(hipStreamSynchronize(st));
  }
  // cal A Z and cal E Z in column chunks of <= 64
  const int CH = 64;
  chunk.alloc((size_t)c->n * CH * 2);
  double* in = chunk.p;
  double* out = chunk.p + (size_t)c->n * CH;
  for (int c0 = 0; c0 < cz; c0 += CH) {
    const int w = std::min(CH, cz - c0);
    launch_copy_cols(st, nv, w, dZ.p, cz, c0, in, w, 0, 1.0);
    launch_spmm(st, nv, c->A.rp.p, c->A.ci.p, c->A.v.p, in, w, nullptr, out, w, nullptr, 0, 1.0, 0.0, nullptr, w);
    if (c->q > 0) {
      c->scratch.ensure((size_t)c->q * w + 64);
      HIPCHK(hipMemsetAsync(c->scratch.p, 0, sizeof(double) * c->q * w, st));
      launch_gemm_tn(st, nv, c->q, w, c->V.p, c->q, in, w, c->scratch.p, w);
      launch_gemm_nn(st, nv, c->q, w, c->U.p, c->q, c->scratch.p, w, out, w, -1.0, 1.0);
    }
    launch_copy_cols(st, nv, w, out, w, 0, S.p, wtot, c0, 1.0);
    launch_spmm(st, nv, c->E.rp.p, c->E.ci.p, c->E.v.p, in, w, nullptr, out, w, nullptr, 0, 1.0, 0.0, nullptr, w);
    launch_copy_cols(st, nv, w, out, w, 0, S.p, wtot, cz + c0, 1.0);
  }
  // project every column: P^T s
  Restore<int> keep_q(c->q);
  for (int c0 = 0; c0 < wtot; c0 += CH) {
    const int w = std::min(CH, wtot - c0);
    launch_copy_cols(st, nv, w, S.p, wtot, c0, in, w, 0, 1.0);
    project_panel(c, in, w);
    launch_copy_cols(st, nv, w, in, w, 0, S.p, wtot, c0, 1.0);
  }
  G.alloc((size_t)wtot * wtot);
  HIPCHK(hipMemsetAsync(G.p, 0, sizeof(double) * wtot * wtot, st));
  launch_gemm_tn(st, nv, wtot, wtot, S.p, wtot, S.p, wtot, G.p, wtot);
  std::vector<double> Gh((size_t)wtot * wtot);
  HIPCHK(hipMemcpyAsync(Gh.data(), G.p, sizeof(double) * wtot * wtot, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  // residual = U S U^T with U = [G, H, Wp], S swaps the first two blocks;
  // ||.||_F^2 = trace(S Gram S Gram)
  auto perm = [&](int i) { return i < cz ? i + cz : (i < 2 * cz ? i - cz : i); };
  double tr = 0.0;
  for (int i = 0; i < wtot; ++i)
    for (int j = 0; j < wtot; ++j)
      tr += Gh[(size_t)perm(i) * wtot + j] * Gh[(size_t)perm(j) * wtot + i];
  *res2_out = tr;
  API_END
}

int ricadi_factor_cols(ricadi_ctx* c, int* c_out) {
  REQUIRE(c && c_out, RICADI_EINVAL, "NULL argument");
  *c_out = c->zc;
  return RICADI_OK;
}

int ricadi_factor_get(ricadi_ctx* c, double* Z_out, int cz) {
  REQUIRE(c && Z_out, RICADI_EINVAL, "NULL argument");
  REQUIRE(cz == c->zc && cz > 0, RICADI_EINVAL, "column count does not match the device factor");
  API_BEGIN
  HIPCHK(hipMemcpy2DAsync(Z_out, sizeof(double) * cz, c->Z.p, sizeof(double) * c->zld,
                          sizeof(double) * cz, c->nv, hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  API_END
}

int ricadi_factor_set(ricadi_ctx* c, const double* Z, int cz) {
  REQUIRE(c && c->has_op, RICADI_ESTATE, "set the operator first");
  REQUIRE(Z && cz > 0, RICADI_EINVAL, "bad argument");
  API_BEGIN
  factor_reserve(c, cz);
  HIPCHK(hipMemcpyAsync(c->Z.p, Z, sizeof(double) * c->nv * cz, hipMemcpyHostToDevice, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  c->zc = cz;
  API_END
}

}  // extern "C"

