/* ricadi.h -- C-ABI of libricadi_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the low-rank Newton-ADI hot path that optconpy reaches
 * through `import sadptprj_riclyap_adi.{proj_ric_utils,lin_alg_utils}`
 * (/root/reference/optcont_main.py:13-14, /root/reference/solve_dae_ric.py:3-4).
 * Every entry point below names the reference-side call it stands behind.
 *
 * Conventions
 *  - plain C types only; all matrices FP64, all indices int32;
 *  - sparse matrices are CSR (rowptr[nrows+1], col[nnz], val[nnz]), sorted or not;
 *  - dense panels are ROW-MAJOR, `ld` = number of columns unless stated
 *    (a C-order numpy array of shape (n, m) is passed as is);
 *  - the caller owns every host buffer; the library copies in / out and keeps
 *    device state only inside the opaque context;
 *  - every function returns 0 on success or a negative RICADI_E* code;
 *    ricadi_last_error() gives the message.  Non-convergence of ADI/Newton is
 *    not an error (the reference just stops at *_max_steps); iteration counts
 *    and histories come back through the stats arrays;
 *  - a context is bound to one GPU and one HIP stream and is NOT thread safe.
 *
 * The operator handled by a context is the saddle-point matrix
 *
 *      S(alpha, beta) = [ beta*A + alpha*E - U*Vt    Jt ]
 *                       [ J                           0  ]
 *
 * with A ("cal A"), E ("cal E") NV x NV, J NP x NV, and an optional low-rank
 * term U (NV x q), Vt (q x NV).  An ADI shift p is (alpha, beta) = (p, 1); the
 * Leray projection uses (1, 0).
 */
#ifndef RICADI_H
#define RICADI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RICADI_OK 0
#define RICADI_EINVAL (-1)    /* bad argument                               */
#define RICADI_EHIP (-2)      /* HIP / rocSOLVER runtime error              */
#define RICADI_ENOCONV (-3)   /* inner Krylov solve missed its tolerance    */
#define RICADI_EBREAKDOWN (-4)/* singular block / coarse matrix             */
#define RICADI_ESTATE (-5)    /* call order violated (operator not set ...) */

typedef struct ricadi_ctx ricadi_ctx;

/* Options of the inner solver (block-Jacobi + coarse-level preconditioned
 * GMRES).  Zero-initialise and call ricadi_default_opts() first.            */
typedef struct ricadi_opts {
  double gmres_tol;      /* relative residual per column (default 1e-10, the
                            unit of work of SURVEY.md section 8d)            */
  int gmres_restart;     /* max Krylov vectors per cycle (default 30); cycles
                            start at 10 vectors and grow to this whenever a
                            cycle gains less than a factor 10 on some column */
  int gmres_maxit;       /* max iterations per solve (default 3000)         */
  int bj_block;          /* block-Jacobi block size, <= 64 (default 32)     */
  int agg_v;             /* velocity aggregate size of the coarse level     */
  int agg_p;             /* pressure aggregate size of the coarse level     */
  int coarse_max;        /* cap on coarse dimension (default 4096); a stiffness-dominated operator
                            (ricadi_host_sa_criterion true) may use 1.5 x this for its dense inverse:
                            see max_levels and ricadi_host_plan_levels        */
  int use_coarse;        /* 0: one-level block-Jacobi only                  */
  int max_levels;        /* 2: two-level method only.  3 (default): where the coarse problem of the
                            aggregates exceeds coarse_max, it is handed to a child level (the same
                            preconditioner on its Galerkin matrices, aggregates = pairs of velocity
                            aggregates + single pressure aggregates, dense inverse <= coarse_max*9/8)
                            instead of growing the aggregates until a dense inverse fits; the
                            aggregates then only grow (x1.5 per step) until that gentle child fits:
                            0.55 k_v + k_p <= coarse_max (n ~ 1e5 keeps agg_v / agg_p).  A harder-
                            coarsened child makes GMRES stagnate.  Operators the
                            smoothed prolongation applies to keep TWO levels instead, up to k <= 1.5
                            coarse_max, and grow their aggregates up to (121, 182) for it: at their small
                            shifts the child is a poor stand-in for the inverse (n = 2e5: 229 vs 118
                            iterations), and the smoothed coarse operator does not go with a child      */
  int verbose;
  int compress_qr;       /* ricadi_compress: 1 (default) = thin block QR + SVD of R, the reference's
                            "QR ... SVD" (singular values resolved to eps*s_1), for factors of up to
                            1024 columns -- wider ones take the Gram route;  0 = always Gram matrix +
                            eigendecomposition (resolves singular values down to sqrt(eps)*s_1)      */
} ricadi_opts;

/* Parameters of the ADI / Newton loops; same meaning as the keys of the
 * reference's `nwtn_adi_dict` (/root/reference/optcont_main.py:122-131).    */
typedef struct ricadi_adi_params {
  int adi_max_steps;
  double adi_newZ_reltol;
  int nwtn_max_steps;
  double nwtn_upd_reltol;
  double nwtn_upd_abstol;
  int project_w;         /* project the rhs factor first (default 1)        */
  int verbose;
  int compress_cols;     /* > 0: recompress the device factor whenever it has
                            grown by this many columns (truncation at
                            sqrt(eps)*sigma_1, i.e. exact to rounding in Z Z^T).
                            0: ricadi_lyap_adi keeps the raw ADI columns like
                            the reference; ricadi_ric_newtonadi uses 512.      */
  int sweep_width;       /* 1 (default): ADI steps one at a time, as the reference.
                            G > 1: sweep form -- G consecutive steps (distinct
                            shifts) are solved together in one batched GMRES and
                            recombined with their Cauchy matrix; same Z Z^T, the
                            stopping rule is applied every G steps.  Falls back to
                            1 when the shift list has repeats, fewer than G
                            entries or an ill-conditioned Cauchy matrix.  G <= 16;
                            the drop-in's proj_alg_ric_newtonadi and the benchmark
                            use 16 (the whole shift cycle of cfg2 in one sweep).  */
} ricadi_adi_params;

const char* ricadi_last_error(void);
int ricadi_version(void);
/* sizeof(ricadi_opts) / sizeof(ricadi_adi_params) as THIS build of the library sees
 * them.  A binding whose mirror of the structs has another size must refuse to load:
 * the library reads every field of the structs it is handed, so a shorter mirror makes
 * it read past the caller's buffer (optconpy_amd/_lib.py checks this at load time). */
int ricadi_sizeof_opts(void);
int ricadi_sizeof_adi_params(void);
/* Field types of the two structs in declaration order, d = double, i = int:
 * "ricadi_opts:<types>;ricadi_adi_params:<types>" -- sizes alone cannot see a
 * trailing int that hides in the tail padding.                                   */
const char* ricadi_struct_signature(void);
void ricadi_default_opts(ricadi_opts* o);
void ricadi_default_adi_params(ricadi_adi_params* p);

/* ---- context ---------------------------------------------------------- */
int ricadi_create(int device_id, ricadi_ctx** ctx);
int ricadi_destroy(ricadi_ctx* ctx);
int ricadi_set_opts(ricadi_ctx* ctx, const ricadi_opts* o);
/* HIP stream all work of this context is enqueued on (as a void*).          */
void* ricadi_stream(ricadi_ctx* ctx);
int ricadi_synchronize(ricadi_ctx* ctx);

/* ---- operator ---------------------------------------------------------
 * Replaces the (amat, mmat, jmat) arguments of
 * pru.solve_proj_lyap_stein / pru.proj_alg_ric_newtonadi
 * (/root/reference/tests/test_units_compfacres_compress.py:62-64,
 *  /root/reference/optcont_main.py:488-492, solve_dae_ric.py:152-159) and of
 * lau.solve_sadpnt_smw (/root/reference/solve_dae_ric.py:192-194).
 * A = cal A, E = cal E already in the orientation they are applied in
 * (the Python shim resolves `transposed`).                                  */
int ricadi_set_operator(ricadi_ctx* ctx, int nv, int np,
                        const int32_t* a_rowptr, const int32_t* a_col, const double* a_val,
                        const int32_t* e_rowptr, const int32_t* e_col, const double* e_val,
                        const int32_t* j_rowptr, const int32_t* j_col, const double* j_val);

/* Drop the per-shift data (assembled values, block inverses, coarse inverse)
 * cached for every (alpha, beta) used so far; they are rebuilt on demand.    */
int ricadi_clear_cache(ricadi_ctx* ctx);

/* Dimension-only context (no operator): enough for ricadi_compress and for
 * ricadi_gain with an explicit `mt_*` matrix, which need NV only.           */
int ricadi_set_dims(ricadi_ctx* ctx, int nv);

/* Low-rank term  - U * Vt ;  U is NV x q row-major, V is NV x q row-major
 * (i.e. Vt = V^T).  q = 0 removes it.  Stands behind the umat / vmat
 * arguments of lau.solve_sadpnt_smw (/root/reference/solve_dae_ric.py:192-194,
 * optcont_main.py:510-514).                                                 */
int ricadi_set_lowrank(ricadi_ctx* ctx, const double* U, const double* V, int q);

/* ---- K1: saddle-point SpMM (testable kernel) --------------------------
 * Y = S(alpha, beta) * X for an n x m panel, n = NV + NP (host buffers).    */
int ricadi_spmm(ricadi_ctx* ctx, double alpha, double beta,
                const double* X, int m, double* Y);

/* Preconditioner alone: Z = P(alpha,beta)^-1 * R  (n x m), for tests.       */
int ricadi_precond_apply(ricadi_ctx* ctx, double alpha, double beta,
                         const double* R, int m, double* Z);

/* ---- one shift-solve ---------------------------------------------------
 * Solve S(alpha,beta) [V; L] = [R; Rp] for an NV x m panel R (Rp may be
 * NULL = 0).  X_out is n x m (velocity rows first).  iters_out[1],
 * relres_out[m] may be NULL.  This is the unit of the `shift-solves/s`
 * metric; stands behind lau.solve_sadpnt_smw.                              */
int ricadi_shift_solve(ricadi_ctx* ctx, double alpha, double beta,
                       const double* R, const double* Rp, int m,
                       double* X_out, int* iters_out, double* relres_out);

/* ---- a2: low-rank ADI for the projected Lyapunov equation -------------
 * pru.solve_proj_lyap_stein (/root/reference/tests/test_units_compfacres_compress.py:62-64).
 * W is NV x m.  Z_out must hold NV x (adi_max_steps*m) doubles, row-major
 * with ld = *c_out on return (columns are packed).  Z_out may be NULL: the
 * factor then stays on the device (see ricadi_factor_*).
 * stats_out (may be NULL, else >= 8 doubles) receives [steps, rel_newZ,
 * total_gmres_iters, shift_solves, ||W_end^T W_end||_F, shift-solves that
 * missed the GMRES tolerance, their worst relative residual].               */
int ricadi_lyap_adi(ricadi_ctx* ctx, const double* shifts, int nshifts,
                    const double* W, int m, const ricadi_adi_params* prm,
                    double* Z_out, int* c_out, double* stats_out);

/* ---- a1: Newton-Kleinman ADI for the projected Riccati equation -------
 * pru.proj_alg_ric_newtonadi (/root/reference/optcont_main.py:488-492,
 * /root/reference/solve_dae_ric.py:152-159).  B is NV x nb dense, W NV x mw,
 * Z0 NV x c0 (or NULL), oldB NV x nb (mtxoldb, or NULL).  Z_out capacity
 * NV x zcap doubles (zcap >= adi_max_steps*(mw+nb)); may be NULL.
 * stats_out (>= 12 doubles): [newton_steps, last_upd_abs, last_upd_rel,
 * total_adi_steps, total_gmres_iters, shift_solves, shift-solves that missed
 * the GMRES tolerance, their worst relative residual, ||W_end^T W_end||_F of the
 * last Lyapunov solve (its projected residual norm: `check_lyap_res`,
 * /root/reference/optcont_main.py:130), ||W_0^T W_0||_F of its right-hand side,
 * solves repeated with wider storage, batched ADI sweeps run (sweep form)].     */
int ricadi_ric_newtonadi(ricadi_ctx* ctx, const double* shifts, int nshifts,
                         const double* B, int nb, const double* W, int mw,
                         const double* Z0, int c0, const double* oldB,
                         const ricadi_adi_params* prm,
                         double* Z_out, int zcap, int* c_out, double* stats_out);

/* ---- a3: column compression -------------------------------------------
 * pru.compress_Zsvd (/root/reference/optcont_main.py:498-499,
 * solve_dae_ric.py:162-163).  Keeps singular values > thresh (thresh < 0:
 * no threshold) and at most kmax (kmax <= 0: no cap).  Z == NULL compresses
 * the factor left on the device by the last ADI / Newton call.  Zc_out is
 * NV x *k_out row-major; sv_out (may be NULL) receives min(c, NV) singular
 * values.                                                                  */
int ricadi_compress(ricadi_ctx* ctx, const double* Z, int c, double thresh, int kmax,
                    double* Zc_out, int* k_out, double* sv_out);

/* The recompression the ADI / Newton drivers apply to their own factor while it grows (no counterpart
 * in the reference, which compresses once through pru.compress_Zsvd,
 * /root/reference/solve_dae_ric.py:161-165): Zc with Zc Zc^T = Z Z^T up to rel^2 ||Z Z^T||, by a pivoted
 * Cholesky factorisation of the Gram matrix and an orthonormalisation of its factor's rows -- no
 * eigensolver, no singular values.  Exposed for tests and timing; `rel` <= 0 takes the drivers' own
 * level (3e-8).  Zc_out must hold NV x c doubles; *k_out columns are written (row-major, ld = *k_out).  */
int ricadi_recompress(ricadi_ctx* ctx, const double* Z, int c, double rel, double* Zc_out, int* k_out);

/* ---- a4: factored product  E * (Z * (Z^T * B)) -------------------------
 * pru.get_mTzzTtb(MT, Z, tb) (/root/reference/optcont_main.py:505-506,
 * solve_dae_ric.py:101,183,189); the feedback gain is its negative.
 * `mt_*` is the CSR of the matrix applied from the left (NV x NV); NULL uses
 * cal E of the context.  Z == NULL uses the device-resident factor.         */
int ricadi_gain(ricadi_ctx* ctx,
                const int32_t* mt_rowptr, const int32_t* mt_col, const double* mt_val,
                const double* Z, int c, const double* B, int nb, double* K_out);

/* ---- a5: squared projected Lyapunov residual norm from factors ---------
 * pru.comp_proj_lyap_res_norm(Z, F, M, W, J)
 * (/root/reference/tests/test_units_compfacres_compress.py:82,104).
 * Uses the context operator: cal A = F^T, cal E = M^T, low-rank term.      */
int ricadi_lyap_res_norm(ricadi_ctx* ctx, const double* Z, int c,
                         const double* W, int m, double* res2_out);

/* ---- device-resident factor -------------------------------------------- */
int ricadi_factor_cols(ricadi_ctx* ctx, int* c_out);
int ricadi_factor_get(ricadi_ctx* ctx, double* Z_out, int c);
int ricadi_factor_set(ricadi_ctx* ctx, const double* Z, int c);
/* ... into a DEVICE buffer (NV x c row-major, ld = c; c = ricadi_factor_cols) */
int ricadi_factor_get_dev(ricadi_ctx* ctx, double* dZ_out, int c);

/* a1 with every panel ALREADY IN HBM (no PCIe traffic inside the call; the reference's callers hand numpy
 * arrays over, /root/reference/solve_dae_ric.py:152-159 -- this is the same call for a host side that keeps
 * its panels on the device between calls, e.g. one backward time step after the other): dB NV x nb, dW NV x mw,
 * dZ0 NV x c0 (NULL with c0 = 0), dOldB NV x nb or NULL, all row-major with leading dimension = width.
 * The new iterate stays in the context's factor (ricadi_factor_cols / _get / _get_dev).
 * stats_out as ricadi_ric_newtonadi.                                                                        */
int ricadi_ric_newtonadi_dev(ricadi_ctx* ctx, const double* shifts, int nshifts,
                             const double* dB, int nb, const double* dW, int mw,
                             const double* dZ0, int c0, const double* dOldB,
                             const ricadi_adi_params* prm, int* c_out, double* stats_out);

/* ---- device-pointer level (multi-GPU orchestration, benchmarks) --------
 * Same operations on buffers that already live in HBM (e.g. torch tensors'
 * data_ptr()); nothing is copied to or from the host.                      */
int ricadi_spmm_dev(ricadi_ctx* ctx, double alpha, double beta,
                    const double* dX, int m, double* dY);
int ricadi_shift_solve_dev(ricadi_ctx* ctx, double alpha, double beta,
                           const double* dR, int m, double* dX,
                           int* iters_out, double* relres_out);
/* The shifts of one ADI sweep in ONE batched solve: for g < ng solve
 *   S(alphas[g], betas[g]) X_g = [R_g; 0]
 * with R_g = dR + g*r_stride (NV x m each; r_stride in doubles, 0 = the same
 * right-hand side for every shift, the shift-parallel sweep) and X_g =
 * dX + g*n*m (n x m each).  All groups advance in lockstep inside one launch
 * sequence (grid.z = groups still iterating), which fills the chip where a
 * single n ~ 3e4 panel cannot.  ng <= 16, ng*m <= 2048.  iters_out: ng ints,
 * relres_out: ng*m doubles (either may be NULL).  Returns RICADI_ENOCONV if a
 * group stopped at gmres_maxit.  Counterpart of the per-shift factorise-and-
 * solve work inside pru.solve_proj_lyap_stein / pru.proj_alg_ric_newtonadi (the package
 * body is not in /root/reference; call sites: /root/reference/solve_dae_ric.py:152-159,
 * /root/reference/optcont_main.py:488-492,
 * /root/reference/tests/test_units_compfacres_compress.py:62-64).                */
int ricadi_shift_solve_batch_dev(ricadi_ctx* ctx, int ng, const double* alphas,
                                 const double* betas, const double* dR,
                                 int64_t r_stride, int m, double* dX,
                                 int* iters_out, double* relres_out);
/* dW (NV x m) += coef * E * dV (first NV rows of an n x m or NV x m panel) */
int ricadi_apply_e_dev(ricadi_ctx* ctx, double coef, const double* dV, int m, double* dW);
/* dOut (nrows x m) = sum_i coef[i] * panel_i, panel_i = dBasis + i*stride
 * (stride in doubles); coef is a host array.  Used by the shift-parallel
 * Cauchy recombination.                              */
int ricadi_lincomb_dev(ricadi_ctx* ctx, int nrows, int m, int nvec, const double* dBasis,
                       int64_t stride, const double* coef, double* dOut);
/* Cauchy recombination of one ADI sweep on the device:
 * dU holds the G solutions U_i (G x NV x m, contiguous, NV rows each); with
 * rinv = R^-1 (G x G row-major, C = R^T R) and cinv1 = C^-1 1 (host arrays from
 * ricadi_host_cauchy):  dZ (NV x G*m, row-major) = U (R^-1 (x) I_m),
 * dW (NV x m) += E U ((C^-1 1) (x) I_m),  *n2_out = ||dZ||_F^2.                */
int ricadi_sweep_recombine_dev(ricadi_ctx* ctx, int G, const double* dU, int m,
                               const double* rinv, const double* cinv1,
                               double* dZ, double* dW, double* n2_out);
/* The same with the solutions in an arbitrary order and with unused slots, as an all-gather
 * of per-rank blocks delivers them: dU holds nslot panels (nslot x NV x m, contiguous);
 * coefz (nslot x G, row-major) and coefw (nslot) are the rows of R^-1 and the entries of
 * C^-1 1 of the shift each slot carries (zero rows for padding slots and for slots that
 * belong to another column part):  dZ (NV x G*m) = sum_i coefz[i][:] (x) U_i,
 * dW (NV x m) += E sum_i coefw[i] U_i.  The data are consumed where the collective put
 * them; only the small coefficient table is permuted.  block_n2_out (G doubles, may be
 * NULL): squared Frobenius norm of every column block of dZ -- with coefz the rows of the
 * upper triangular R^-1 these are the blocks of the step-by-step iteration, which is what
 * the reference's stopping rule looks at.                                               */
int ricadi_sweep_recombine_slots_dev(ricadi_ctx* ctx, int nslot, int G, const double* dU, int m,
                                     const double* coefz, const double* coefw,
                                     double* dZ, double* dW, double* n2_out, double* block_n2_out);
/* dK (NV x nb) = coef * E * (Z * (Z^T B)) for a device-resident factor dZ
 * (NV x c, row-major with leading dimension ldz) and dB (NV x nb).          */
int ricadi_gain_dev(ricadi_ctx* ctx, double coef, const double* dZ, int c, int ldz,
                    const double* dB, int nb, double* dK);
/* Frobenius norms of the m columns' Gram matrix: out = ||W^T W||_F, and the
 * squared F-norm of the panel in nrm2 (both may be NULL).                  */
int ricadi_panel_norms_dev(ricadi_ctx* ctx, const double* dW, int nrows, int m,
                           double* gram_fro, double* nrm2);
/* Average duration (ms) of the last timed kernel class, measured with HIP
 * events on the context stream: which = 0 SpMM.                            */
int ricadi_time_spmm_dev(ricadi_ctx* ctx, double alpha, double beta, const double* dX,
                         int m, double* dY, int reps, double* ms_per_launch);
/* The same for the batched launch of the hot path: ng panels (dX + g*n*m ->
 * dY + g*n*m), shift g = (alphas[g], betas[g]); one launch covers all of them. */
int ricadi_time_spmm_batch_dev(ricadi_ctx* ctx, int ng, const double* alphas,
                               const double* betas, const double* dX, int m,
                               double* dY, int reps, double* ms_per_launch);

/* One launch of a hot-path kernel class exactly as the batched GMRES issues it (ng groups,
 * panel width m, nvec Krylov vectors for the Arnoldi classes), on the solver's own
 * workspace buffers; average over `reps` launches, HIP events on the context stream.
 * Behind the per-kernel roofline objects of bench.py.  No reference counterpart.      */
#define RICADI_TK_SPMM 0          /* tile SpMM of the saddle operator                     */
#define RICADI_TK_BLOCK_V 1       /* block-Jacobi sweep, velocity blocks                  */
#define RICADI_TK_BLOCK_P 2       /* block-Jacobi sweep, Schur (pressure) blocks          */
#define RICADI_TK_COARSE 3        /* dense coarse apply                                   */
#define RICADI_TK_SPMM_SY 4       /* tile SpMM of the prolongated operator S*Y            */
#define RICADI_TK_DOTS 5          /* cols_dots + reduce_partials                          */
#define RICADI_TK_UPDATE_DOTS 6   /* cols_update_dots + reduce_partials                   */
#define RICADI_TK_UPDATE 7        /* cols_update (writes the new Krylov vector)           */
#define RICADI_TK_PRECOND 8       /* the whole preconditioner application (all launches)  */
#define RICADI_TK_RESTRICT 9      /* restriction Y^T r (CSR SpMM with unit values)        */
/* 10 + k: stage k of the preconditioner application ALONE, issued by the solver's own code path (so the
 * kernel and its template instance are the ones the iteration launches at this size):
 * 0 restriction, 1 coarse apply (dense inverse or child cycle), 2 pressure rows of r - (S Y) e,
 * 3 first velocity sweep (two-term block sweep with the coarse residual folded in), 4 J product,
 * 5 Schur-complement sweep, 6 last velocity sweep (rectangle sweep with prolongation)             */
#define RICADI_TK_PC_STAGE0 10
int ricadi_time_kernel_dev(ricadi_ctx* ctx, int which, int ng, const double* alphas,
                           const double* betas, int m, int nvec, int reps,
                           double* ms_per_launch);

/* K5: thin QR factorisation Z = Q R of an NV x c host matrix (c <= NV) by block
 * Gram-Schmidt with re-orthogonalisation over 32-column panels, each panel
 * factorised by a Householder TSQR tree.  R_out: c x c row-major upper triangular;
 * Q_out: NV x c or NULL.  The "QR" of the reference's compress_Zsvd comment
 * (/root/reference/optcont_main.py:133-134) and of the Newton update norm.      */
int ricadi_qr(ricadi_ctx* ctx, const double* Z, int c, double* Q_out, double* R_out);

/* Structure of the preconditioner the context built for its operator:
 * out = [NV, NP, velocity blocks, Schur blocks, block size, coarse dimension,
 *        SpMM row blocks, max distinct columns per row block,
 *        preconditioner levels in use, dimension of the dense inverse on the last level,
 *        1 if the GMRES iteration hands the preconditioner the FP16-stored vector,
 *        padded width of the dense rectangles of the last velocity sweep (0: not in that form),
 *        the same for the first (two-term) velocity sweep, NP, nnz(J), nnz(S*Y)];
 *        entries of the restriction,
 *        route the last batch of dense coarse inverses took (0 block Gauss-Jordan without pivoting, 1 rocSOLVER
 *        with partial pivoting; -1 none yet),
 *        kernel of the last saddle SpMM launch (0 CSR, 1 LDS-tiled per group, 2 LDS-tiled multi-shift; +4 with
 *        FP32 x input; -1 none yet),
 *        1 if the last preconditioner application kept the velocity part between its sweeps as an FP32 panel
 *        (first sweep -> pressure step -> last sweep), 0 for an FP64 panel; -1 none yet,
 *        1 if the operator launch of the last iteration (or timing call) wrote w = S z as an FP32 panel for the
 *        Arnoldi passes, 0 for FP64; -1 none yet];
 * nout >= 8; entries beyond nout are not written.                                   */
int ricadi_setup_info(ricadi_ctx* ctx, int* out, int nout);

/* In-place inverses of nb dense k x k matrices (row major, back to back in A, host memory) by the routine the
 * setup uses for the coarse matrices of a batch of shifts: block Gauss-Jordan without pivoting, and -- when a
 * pivot vanishes relative to the scale of its diagonal block -- rocSOLVER's getrf / getri with partial
 * pivoting for the whole batch.  *route_out: the route that finished (as in ricadi_setup_info).  Exported for
 * tests (the production operators never leave route 0).  No reference counterpart (its LU is SuperLU's). */
int ricadi_dense_inverse_batch(ricadi_ctx* ctx, int k, int nb, double* A, int* route_out);

/* Average duration (ms) of the thin QR of a device-resident NV x c factor (the device
 * part of ricadi_qr: Householder TSQR panels inside a block Gram-Schmidt on the MFMA
 * GEMMs), HIP events on the context stream: the TSQR MFMA-utilisation figure.          */
int ricadi_time_qr_dev(ricadi_ctx* ctx, const double* dZ, int c, int reps, double* ms_per_call);

/* Average duration (ms) of the FP64-MFMA Gram kernel G = Z^T Z (the 2*NV*c^2 flop
 * part of ricadi_compress) for a device-resident NV x c factor, HIP events on the
 * context stream.  dG must hold c*c doubles.                                   */
int ricadi_time_gram_dev(ricadi_ctx* ctx, const double* dZ, int c, double* dG, int reps,
                         double* ms_per_launch);

/* ---- recycling of solved right-hand sides -------------------------------
 * depth > 0: every batched solve whose groups share ONE right-hand side panel (the sweeps
 * of the ADI) starts from the least-squares combination of the last `depth` panels it has
 * already solved for the same shifts -- x0_g = sum_e Y_{g,e} C_e with C = argmin ||b - B C||_F
 * over the stored right-hand sides B -- instead of from zero, and stores its own (b, y_g)
 * afterwards.  The residual factors of consecutive ADI sweeps span nearly the same space
 * (cfg2: ||b - B C|| / ||b|| = 5e-2 ... 7e-4 per column from the fourth sweep on), which
 * saves that many digits of every later solve; the tolerance stays relative to ||b||.
 * The ADI / Newton drivers switch it on for their own sweeps (depth 5, 3 beyond n = 2e5;
 * RICADI_RECYCLE=d overrides, 0 = off); this call sets the depth for direct ricadi_shift_solve*_dev calls
 * (default 0, so that repeated identical solves measure what they seem to measure).
 * ricadi_clear_cache() drops the stored panels.  No reference counterpart (SuperLU is direct). */
int ricadi_set_recycle(ricadi_ctx* ctx, int depth);

/* ---- shift-parallel sweeps across processes (one process per GPU) -------
 * The ADI sweeps of ricadi_lyap_adi / ricadi_ric_newtonadi (sweep_width > 1) shard by shift:
 * every rank owns a fixed subset of the shift list (ricadi_host_deal), sets up and solves only
 * its own shifts of a sweep in one batched solve, and the solution panels are exchanged by ONE
 * all-gather per sweep; recombination, recompression, update norm and gain are replicated.  The
 * stopping decisions inside a sweep are taken by every rank on its own from the gathered panels (the
 * block norms are summed in a fixed order, so the ranks see the same bits); a rank whose own setup or
 * solves fail still takes part in the all-gather with a status word set, and all ranks return the error
 * together.  Two transports: RCCL inside the library (ricadi_set_exchange_rccl, below), or -- this call --
 * a collective the host supplies as a callback on two device buffers it owns (the Python binding uses it
 * for gloo groups: CPU tests and the one-GPU rehearsal):
 *   fn(user, send_dev, recv_dev, bytes_per_rank) must place rank r's first bytes_per_rank bytes
 *   of send_dev at recv_dev + r * bytes_per_rank on EVERY rank and return 0 once the data are
 *   visible to work enqueued afterwards on any stream (the library has synchronised its own
 *   stream before the call).
 * send_dev holds send_capacity bytes, recv_dev world * send_capacity; the last 4096 bytes of send_dev
 * (and the last world * 4096 of recv_dev) carry the small control messages, so fn is also called with
 * pointers INSIDE the two buffers and must honour them.  world = 1 (or fn NULL) removes the exchange.  All ranks must make the same sequence of solver calls with the same
 * arguments.  SURVEY.md section 8e; the reference has nothing distributed
 * (/root/reference/solve_dae_ric.py:122 and optcont_main.py:577 are sequential loops).      */
typedef int (*ricadi_allgather_fn)(void* user, const void* send_dev, void* recv_dev,
                                   int64_t bytes_per_rank);
int ricadi_set_exchange(ricadi_ctx* ctx, int rank, int world, ricadi_allgather_fn fn, void* user,
                        void* send_dev, void* recv_dev, int64_t send_capacity);

/* The same sharding with the collective INSIDE the library: RCCL's ncclAllGather (double, count = n * m per
 * solution panel and rank) enqueued on the context's stream between the solves and the recombination -- no host
 * synchronisation, no callback (section 5: the blocks exchanged are the V_i of
 * /root/reference/solve_dae_ric.py:152-159, the reference itself has nothing distributed).
 *   ricadi_rccl_unique_id: 128 bytes from ncclGetUniqueId; rank 0 calls it and hands the bytes to the other
 *     ranks by whatever means the host has (the Python binding: torch.distributed.broadcast_object_list).
 *   ricadi_set_exchange_rccl: joins the communicator of `world` ranks identified by unique_id
 *     (ncclCommInitRank, collective over the ranks; destroyed with the context) -- or, with comm != NULL, uses an
 *     ncclComm_t the caller created (not destroyed by the library).  The library allocates its own send buffer
 *     of send_capacity bytes and receive buffer of world * send_capacity bytes (capacity needed:
 *     max panels per rank and sweep * n * m * 8 + 4096).  world = 1 is allowed and still runs the exchange
 *     path (one rank: the transport test).  unique_id = comm = NULL on a context that already holds a
 *     communicator of the same rank / world keeps it and only re-sizes the buffers.
 *     Replaces a callback set by ricadi_set_exchange and vice versa;
 *     ricadi_set_exchange(ctx, 0, 1, NULL, ...) removes either.
 *   ricadi_exchange_count: collectives the context has issued so far, control messages included (a sweep
 *     costs exactly one; a Newton step one more for the update-norm decision, an ADI call one for statistics). */
int ricadi_rccl_unique_id(void* id_out, int bytes);
int ricadi_set_exchange_rccl(ricadi_ctx* ctx, int rank, int world, const void* unique_id, void* comm,
                             int64_t send_capacity);
int ricadi_exchange_count(ricadi_ctx* ctx, int64_t* count_out);

/* ---- host-side logic exported for CPU tests ---------------------------- */
/* Owner rank of every entry of an ADI shift list under `world` ranks: longest-processing-time
 * dealing on the lockstep cost model  T(rank) = a max_g it_g + b sum_g it_g  (a / b = 3.2: the
 * latency floor and the per-group slope of one lockstep iteration) with
 * predicted iteration counts it(p) that fall with |p| (the small shifts are the slow solves).
 * Deterministic in (shifts, world): every rank computes the same table.  Returns 0 or <0.   */
int ricadi_host_deal(const double* shifts, int nshifts, int world, int32_t* owner_out);
/* Whether the setup smooths the velocity aggregates of the two-level preconditioner for the operator cal A (CSR,
 * NV x NV):  *on_out = 1 iff the row sums of sym(cal A) stay below 0.15 of its diagonal (stiffness-like, not
 * mass-like) and sum |skew part| / sum |off-diagonal symmetric part| <= 0.7 (not convection dominated).  The two
 * ratios are returned (skew ratio -1 when the first test failed).  Host only; no GPU needed.                       */
int ricadi_host_sa_criterion(int nv, const int32_t* a_rowptr, const int32_t* a_col, const double* a_val,
                             double* rowsum_ratio_out, double* skew_ratio_out, int* on_out);
/* The preconditioner hierarchy ricadi_set_operator would choose for (cal A, cal E, J) with these options, computed on
 * the host (before the device-side fallbacks):  out[5] = { levels (1: block-Jacobi only, 2: dense coarse inverse,
 * 3: child level), coarse dimension k, its velocity / pressure parts, smoothed prolongation (0/1) }.  A stiffness-
 * dominated operator (ricadi_host_sa_criterion) keeps two levels up to k <= 1.5 coarse_max and grows its aggregates
 * up to (121, 182) for it; any other goes to a child level as soon as the gentle child fits coarse_max.             */
int ricadi_host_plan_levels(int nv, int np, const int32_t* a_rowptr, const int32_t* a_col, const double* a_val,
                            const int32_t* e_rowptr, const int32_t* e_col, const double* e_val,
                            const int32_t* j_rowptr, const int32_t* j_col, const double* j_val,
                            const ricadi_opts* opts, int32_t* out);
/* Greedy BFS aggregation of the graph of a CSR pattern into blocks of at
 * most bsize rows; blk_out[n]; returns the number of blocks (or <0).        */
int ricadi_host_aggregate(int n, const int32_t* rowptr, const int32_t* col,
                          int bsize, int32_t* blk_out);
/* Cauchy recombination data for a sweep of g distinct real negative shifts
 * (SURVEY.md section 8e):  C_ij = -1/(p_i+p_j) = R^T R (upper R);
 * rinv_out = R^-1 (g x g row-major), cinv1_out = C^-1 * ones (g), both in closed form (partial
 * fractions of the ADI steps' rational functions: accurate to a few ulp per entry whatever the
 * condition of C).  RICADI_EBREAKDOWN for repeated shifts and for sweeps whose recombination would
 * amplify the solves' errors by more than ~1e5 (smallest relative pivot of C below 1e-6: shifts too
 * many / too close); the drivers then halve the sweep width.                                    */
int ricadi_host_cauchy(const double* shifts, int g, double* rinv_out, double* cinv1_out);

#ifdef __cplusplus
}
#endif
#endif /* RICADI_H */
