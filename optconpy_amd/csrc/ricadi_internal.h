// ricadi_internal.h -- shared declarations of libricadi_hip.so (not installed).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ricadi.h"

#define RICADI_MAX_M 128
#define RICADI_MAX_GROUPS 16

namespace ricadi {

// ---- host-side CSR helper --------------------------------------------------
struct HostCsr {
  int nrows = 0, ncols = 0;
  std::vector<int> rp, ci;
  std::vector<double> v;
  size_t nnz() const { return ci.size(); }
};
HostCsr make_csr(int nrows, int ncols, const int32_t* rp, const int32_t* ci, const double* v);
HostCsr transpose(const HostCsr& a);
void sort_rows(HostCsr& a);

int aggregate(int n, const int* rp, const int* ci, int bsize, int* blk);

// Host-built description of the saddle operator and the two-level
// preconditioner's fixed (shift independent) parts.
struct HostSetup {
  int nv = 0, np = 0, n = 0;
  // unified saddle pattern, three value sources
  std::vector<int> s_rp, s_ci;
  std::vector<double> s_srcA, s_srcE, s_srcJ;
  // diagonals of A and E
  std::vector<double> dA, dE;
  // block-Jacobi, velocity block
  int bs = 32;
  int nbv = 0;
  std::vector<int> bv_ptr, bv_rows;
  std::vector<double> bv_A, bv_E;  // nbv x bs x bs
  // block-Jacobi, Schur complement
  int nbp = 0;
  std::vector<int> bp_ptr, bp_rows;  // rows are pressure-local (0..np)
  // dense bs x bs blocks of J for every (pressure block, coupled velocity block) pair:
  // row = pressure dof local to its block, column = velocity dof local to its block
  std::vector<int> jd_ptr, jd_vblk;
  std::vector<double> jd_val;
  // LDS-tiled SpMM: rows grouped in blocks of <= 64 (pairs of block-Jacobi
  // aggregates = compact mesh patches); per block the distinct columns and
  // 16-bit local column indices; arrays in block order
  int sb_nblk = 0, sb_max_cols = 0, sb_max_nnz = 0;
  std::vector<int> sb_rowptr, sb_rows, sb_rp, sb_cptr, sb_cols, sb_perm;
  std::vector<uint16_t> sb_lidx;
  // coarse level
  int kc = 0, kcv = 0, kcp = 0;
  std::vector<int> agg_ptr, agg_rows;  // aggregates over all n dofs
  std::vector<int> aggof;              // n -> coarse index
  std::vector<double> E0, EM, EJ;      // kc x kc dense
  // Multilevel: the coarse saddle problem is NOT inverted densely (kc > coarse_max with the base
  // aggregate sizes); its Galerkin matrices Yv^T A Yv, Yv^T E Yv, Yp^T J Yv go to a child level.
  bool multilevel = false;
  HostCsr l1A, l1E, l1J;
  // Smoothed aggregation (round 3; two-level setups only): the velocity part of the prolongation is
  //   P_v = (I - omega D^-1 K0) Y_v,  K0 = sym(cal A), D = diag(K0)
  // -- shift independent, so that P^T S(p) P and S(p) P stay linear in (alpha, beta).  p_*: P by rows (all n rows;
  // pressure rows are the piecewise-constant ones), pt_*: P^T by rows (the restriction), pd_*: P - Y on the
  // velocity rows (its action on the coarse vector is folded into the first velocity sweep).  With sa == false
  // P = Y and these arrays are empty.
  bool sa = false;
  std::vector<int> p_rp, p_ci, pt_rp, pt_ci, pd_rp, pd_ci;
  std::vector<double> p_v, pt_v, pd_v;
  // S * Y: CSR over (row, aggregate) with the three value sources of the saddle pattern
  std::vector<int> sy_rp, sy_ci;
  std::vector<double> sy_A, sy_E, sy_J;
  // ... and in tile format on the row blocks of the saddle operator
  int syb_max_cols = 0;
  std::vector<int> syb_rp, syb_cptr, syb_cols, syb_perm;
  std::vector<uint16_t> syb_lidx;
};
void build_setup(const HostCsr& A, const HostCsr& E, const HostCsr& J, const ricadi_opts& o,
                 HostSetup& hs, int max_levels = 2, double sa_omega = 0.0);
bool sa_criterion(const HostCsr& A, double& rs_out, double& gamma_out);
int cauchy_data(const double* shifts, int g, double* rinv, double* cinv1);
int deal_shifts(const double* shifts, int ns, int world, int32_t* owner);
int gram_lstsq(int h, int m, const double* Ghh, const double* Ghb, double rtol, double* Y);
// the same on the column-normalised problem (unit diagonal), solution scaled back; Ghh / Ghb are overwritten
int gram_lstsq_scaled(int h, int m, std::vector<double>& Ghh, std::vector<double>& Ghb, double rtol,
                      std::vector<double>& Y);

// ---- batches of panels ------------------------------------------------------------
// A *batch* is a set of up to RICADI_MAX_GROUPS panels (one per ADI shift of a
// sweep), stored group-major: panel g of a buffer starts g * (group stride)
// doubles behind panel 0.  Batched launches add the ACTIVE groups as grid.z:
// block z works on group gid[z], so groups whose solve has converged simply
// drop out of the table while their data stay where they are.  Shift-dependent
// operands (matrix values, block inverses, coarse inverse) come as one pointer
// per group id.  Both tables travel by value in the kernel arguments.
struct GroupTab {
  int ng;
  int gid[RICADI_MAX_GROUPS];
};
template <class T>
struct GroupPtrsT {
  const T* p[RICADI_MAX_GROUPS];
};
typedef GroupPtrsT<double> GroupPtrs;
typedef GroupPtrsT<float> GroupPtrsF;   // FP32-stored preconditioner operands
typedef GroupPtrsT<uint16_t> GroupPtrsH;   // BF16-stored block operands of the sweeps (bit patterns; round 4)
struct GroupInts {            // one small integer per group id (by value)
  int v[RICADI_MAX_GROUPS];
};
inline GroupInts same_int(int k) {
  GroupInts g;
  for (int i = 0; i < RICADI_MAX_GROUPS; ++i) g.v[i] = k;
  return g;
}
inline GroupTab single_group() {
  GroupTab t{};
  t.ng = 1;
  return t;
}
template <class T>
inline GroupPtrsT<T> same_ptr(const T* q) {
  GroupPtrsT<T> g;
  for (int i = 0; i < RICADI_MAX_GROUPS; ++i) g.p[i] = q;
  return g;
}

// Coarse-level prolongation fused into a block-Jacobi sweep: every row the sweep
// writes gets  + ec[aggof[row], :]  (ec: kc x m per group, stride gse), and surplus
// waves do the same for rows [row0, row0 + nextra) outside the blocks.
struct ProlongArgs {
  const int* aggof = nullptr;
  const double* ec = nullptr;
  size_t gse = 0;
  int row0 = 0, nextra = 0;
  double* out2 = nullptr;   // if set: the sweep's result before the coarse part is added
  size_t gs2 = 0;
  float* out32 = nullptr;   // if set: FP32 copy of what the sweep writes (flexible GMRES keeps Z_j = P^-1 v_j)
  size_t gs32 = 0;
  int only32 = 0;           // with out32: the FP64 result is not stored (the operator reads the FP32 copy)
  int old32 = 0;            // rectangle sweep: the rows it updates are read from out32 (FP32 intermediate of the cycle)
  // Fixed-stride record per 32-row block (ricadi_ctx::sw_meta): {nb, ni of the rectangle sweep, ni of the two-term
  // sweep, 0 | rows[32] | aggregate of every row [32] | input rows of the rectangle sweep | of the two-term sweep},
  // lists padded with their last entry.  With it a wave has every index after ONE load round (block pointers ->
  // row lists -> aggregate map were three dependent ones); bm_in = offset of the launched sweep's input list.
  const int* bmeta = nullptr;
  int bm_stride = 0, bm_in = 0, bm_ni = 0;   // bm_ni: which header word holds ni (1 or 2)
};

// Low-rank term fused into an SpMM epilogue:  y[row, :] -= U[row, :] * c  for
// row < nrows, with c = V^T x (q x m per group, stride gsc) reduced beforehand.
struct LowRankArgs {
  const double* U = nullptr;
  const double* c = nullptr;
  size_t gsc = 0;
  int q = 0, nrows = 0;
};

// Input of a block-Jacobi sweep produced on the fly:
//   in[row, :] = base[row, :] + scale * (C * src)[row, :]
// with a CSR matrix C (rows = the sweep's rows; values per group), a panel src (group
// stride gss, same leading dimension as `in` would have) and an optional panel base
// (group stride gsb; NULL = 0).  Saves writing and re-reading the intermediate panel:
// the J^T product of the SIMPLE sweep (base = NULL, scale = 1) and the residual after
// the coarse correction r - (S Y) e (base = r, scale = -1).
struct CsrInArgs {
  const int* rp = nullptr;
  const int* ci = nullptr;
  GroupPtrs v = {};
  const double* src = nullptr;
  size_t gss = 0;
  const double* base = nullptr;
  size_t gsb = 0;
  double scale = 1.0;
};

// One input segment of the two-term block sweep (block_apply2_kernel):
//   iptr == NULL: the block's own rows; moff == NULL: matrix of block b at b * BS * kstride;
//   kstride == 0: row stride = the block's input count.
struct Seg2 {
  const int* iptr = nullptr;
  const int* irows = nullptr;
  const int* moff = nullptr;
  int kstride = 0;
  const double* in = nullptr;
  size_t gs = 0;
  const _Float16* in16 = nullptr;   // segment 1 only: read the input rows from an FP16 panel instead of `in`
};

// y = A x for a CSR matrix with LONG rows (the restriction: one row per aggregate), one wave per row
// (spmm_rowwave_kernel); x FP64 or FP16-stored (x16), panels of m = 16 columns (leading dimension 16), group strides
// gsx / gsy.  spmm_rowwave_pays: rows long enough on average (>= 32 entries) for it to beat the 16-lane-per-row kernel.
bool spmm_rowwave_pays(int nrows, size_t nnz);
void launch_spmm_rowwave(hipStream_t st, const GroupTab& gt, int nrows, const int* rp, const int* ci,
                         const GroupPtrs& vals, const double* x, const _Float16* x16, size_t gsx, double* y, size_t gsy,
                         int m);

// ---- kernel launchers (ricadi_kernels.hip) ---------------------------------
// The *_b launchers are the batched forms (GroupTab + group strides `gs*`, in
// doubles); the plain ones run a single panel.
void launch_spmm_h(hipStream_t st, const GroupTab& gt, int nrows, const int* rp, const int* ci,
                   const GroupPtrs& vals, const double* x, const _Float16* x16, int ldx, size_t gsx, double* y, int ldy,
                   size_t gsy, const _Float16* r16, int ldr, size_t gsr, double alpha, double beta_r, int m, int chunk);
void launch_spmm_b(hipStream_t st, const GroupTab& gt, int nrows, const int* rp, const int* ci,
                   const GroupPtrs& vals, const double* x, int ldx, size_t gsx, const int* xmap,
                   double* y, int ldy, size_t gsy, const double* r, int ldr, size_t gsr,
                   double alpha, double beta_r, int m, const LowRankArgs& lr = LowRankArgs(),
                   int chunk = 16);   // chunk = 8: matrices with short rows (<= ~10 entries)
void launch_spmm_blocked_b(hipStream_t st, const GroupTab& gt, int nblk, const int* rows2,
                           const int* rp2, const int* cols2, const uint16_t* lidx,
                           const GroupPtrs& vals, const double* x, int ldx, size_t gsx, double* y,
                           int ldy, size_t gsy, const double* r, int ldr, size_t gsr, double alpha,
                           double beta_r, int m, int max_cols,
                           const LowRankArgs& lr = LowRankArgs());
// multi-shift form: one read of the value arrays (tile order; vAJ = A part + J part, vE) for
// all active groups; lidx carries the velocity-velocity flag in bit 15; alphas / betas:
// RICADI_MAX_GROUPS coefficients indexed by group id
bool spmm_blocked_ms_ok(int m, int max_cols, size_t panel_rows);
// FP32 input panel: with m == 16 the kernel fills its tile with 16-byte loads and addresses x rows as 16 packed floats
// -- the panel must be packed (ldx == 16, 16-byte aligned); the one caller (saddle_spmm) passes ldx = m
void launch_spmm_blocked_x32(hipStream_t st, const GroupTab& gt, int nblk, const int* rows2, const int* rp2,
                             const int* cols2, const uint16_t* lidx, const GroupPtrs& vals, const float* x, int ldx,
                             size_t gsx, double* y, int ldy, size_t gsy, double alpha, int m, int max_cols, float* y32 = nullptr);
void launch_spmm_blocked_ms_x32(hipStream_t st, const GroupTab& gt, const double* alphas, const double* betas,
                                int nblk, const int* rows2, const int* rp2, const int* cols2, const uint16_t* lidx,
                                const double* vAJ, const double* vE, const float* x, int ldx, size_t gsx, double* y,
                                int ldy, size_t gsy, double alpha, int m, int max_cols, float* y32 = nullptr);
void launch_spmm_blocked_ms(hipStream_t st, const GroupTab& gt, const double* alphas, const double* betas,
                            int nblk, const int* rows2, const int* rp2, const int* cols2,
                            const uint16_t* lidx, const double* vAJ, const double* vE,
                            const double* x, int ldx, size_t gsx, double* y, int ldy, size_t gsy,
                            const double* r, int ldr, size_t gsr, double alpha, double beta_r, int m,
                            int max_cols);
void launch_axpby_b(hipStream_t st, const GroupTab& gt, size_t n, double a, const double* x,
                    size_t gsx, double b, double* y, size_t gsy);
void launch_colscale_b(hipStream_t st, const GroupTab& gt, size_t nrows, int m, const double* a,
                       const double* x, size_t gsx, double b, double* y, size_t gsy,
                       float* yf = nullptr, size_t gsf = 0);
void launch_colscale_b(hipStream_t st, const GroupTab& gt, size_t nrows, int m, const double* a,
                       const double* x, size_t gsx, double b, double* y, size_t gsy, _Float16* yf,
                       size_t gsf);
void launch_cols_dots_b(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                        const _Float16* basis, size_t vstride, size_t gsb, const double* w,
                        size_t gsw, int want_self, double* partial, size_t gsp, double* out,
                        size_t gso);
void launch_cols_update_dots_b(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                               const _Float16* basis, size_t vstride, size_t gsb, const double* h,
                               size_t gsh, double* w, size_t gsw, double* partial, size_t gsp,
                               double* out, size_t gso);
void launch_cols_update_b(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                          const _Float16* basis, size_t vstride, size_t gsb, const double* h,
                          size_t gsh, double sign, const double* w, size_t gsw, const double* scale,
                          double* out, size_t gso, _Float16* outf, size_t gsf);
void launch_cols_update_bk(hipStream_t st, const GroupTab& gt, int nrows, int m, const GroupInts& nvec,
                           const _Float16* basis, size_t vstride, size_t gsb, const double* h, size_t gsh,
                           double* out, size_t gso, const double* acc = nullptr, size_t gsa = 0);
// FP32-stored Krylov basis (arithmetic stays FP64): overloads reading `const float* basis`
void launch_cols_dots_b(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                        const float* basis, size_t vstride, size_t gsb, const double* w, size_t gsw,
                        int want_self, double* partial, size_t gsp, double* out, size_t gso);
void launch_cols_update_dots_b(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                               const float* basis, size_t vstride, size_t gsb, const double* h,
                               size_t gsh, double* w, size_t gsw, double* partial, size_t gsp,
                               double* out, size_t gso);
void launch_cols_update_b(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                          const float* basis, size_t vstride, size_t gsb, const double* h,
                          size_t gsh, double sign, const double* w, size_t gsw, const double* scale,
                          double* out, size_t gso, float* outf, size_t gsf);
void launch_cols_dots_b(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                        const double* basis, size_t vstride, size_t gsb, const double* w, size_t gsw,
                        int want_self, double* partial, size_t gsp, double* out, size_t gso);
void launch_cols_update_dots_b(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                               const double* basis, size_t vstride, size_t gsb, const double* h,
                               size_t gsh, double* w, size_t gsw, double* partial, size_t gsp,
                               double* out, size_t gso);
void launch_cols_update_b(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                          const double* basis, size_t vstride, size_t gsb, const double* h,
                          size_t gsh, double sign, const double* w, size_t gsw, const double* scale,
                          double* out, size_t gso);
void launch_gmres_hess_b(hipStream_t st, const GroupTab& gt, int m, int j, int restart,
                         const double* h1, const double* h2, double* H, double* cs, double* sn,
                         double* g, double* scale, double* resid, const double* bnorm, double tol,
                         double* host_resid = nullptr, double* zero_h1 = nullptr, double* zero_h2 = nullptr,
                         double* hsum = nullptr);
// 16-column FP16 Arnoldi path: the update+dots launch may leave w as it was BEFORE the first projection (no 8-byte
// store per element); the Hessenberg kernel then writes h1 + h2 to `hsum` and the final update uses those on w
bool update_dots_keeps_w(int m, bool fp16_basis, int nvec_max);
void set_update_dots_nostore(bool v);
void launch_gmres_backsolve_b(hipStream_t st, const GroupTab& gt, int m, const GroupInts& k,
                              int restart, const double* H, const double* g, double* y);
void launch_cols_update_bk(hipStream_t st, const GroupTab& gt, int nrows, int m, const GroupInts& nvec,
                           const double* basis, size_t vstride, size_t gsb, const double* h, size_t gsh,
                           double* out, size_t gso, const double* acc = nullptr, size_t gsa = 0);
void launch_cols_update_bk(hipStream_t st, const GroupTab& gt, int nrows, int m, const GroupInts& nvec,
                           const float* basis, size_t vstride, size_t gsb, const double* h, size_t gsh,
                           double* out, size_t gso, const double* acc = nullptr, size_t gsa = 0);
void launch_gmres_start_b(hipStream_t st, const GroupTab& gt, int m, int restart,
                          const double* nrm2, double* g, double* scale, double* resid);
void launch_block_apply_b(hipStream_t st, const GroupTab& gt, int bs, int nblocks, const int* bptr,
                          const int* rows, const GroupPtrs& inv, const double* in, int ldi,
                          size_t gsi, double* out, int ldo, size_t gso, int m, int subtract,
                          const ProlongArgs& pa = ProlongArgs(), const CsrInArgs& ci = CsrInArgs());
void launch_dense_apply_b(hipStream_t st, const GroupTab& gt, int k, int m, const GroupPtrs& Einv,
                          const double* rc, double* ec);
// FP32-stored inverses (leading dimension ldf = k rounded up to 4; bs x bs blocks)
// f32_matrix_cores: the product on v_mfma_f32_16x16x4_f32 (coarse residual rounded to FP32, FP32 accumulation per K
// slice) instead of v_mfma_f64_16x16x4_f64 -- experimental, see the kernel
void launch_dense_apply_b(hipStream_t st, const GroupTab& gt, int k, int m, const GroupPtrsF& Einv,
                          int ldf, const double* rc, double* ec, bool f32_matrix_cores = false);
void launch_block_apply_b(hipStream_t st, const GroupTab& gt, int bs, int nblocks, const int* bptr,
                          const int* rows, const GroupPtrsF& inv, const double* in, int ldi,
                          size_t gsi, double* out, int ldo, size_t gso, int m, int subtract,
                          const ProlongArgs& pa = ProlongArgs(), const CsrInArgs& ci = CsrInArgs());
// rectangular block sweep  out[rows_b] (-)= mats[b] (bs x ks) * in[irows_b]  (+ fused prolongation)
bool block_apply_rect_ok(int bs, int ks);
void launch_block_apply_rect_b(hipStream_t st, const GroupTab& gt, int bs, int ks, int nblocks,
                               const int* bptr, const int* rows, const int* iptr, const int* irows,
                               const GroupPtrs& mats, const double* in, int ldi, size_t gsi, double* out,
                               int ldo, size_t gso, int m, int subtract, const ProlongArgs& pa);
void launch_block_apply_rect_b(hipStream_t st, const GroupTab& gt, int bs, int ks, int nblocks,
                               const int* bptr, const int* rows, const int* iptr, const int* irows,
                               const GroupPtrsF& mats, const double* in, int ldi, size_t gsi, double* out,
                               int ldo, size_t gso, int m, int subtract, const ProlongArgs& pa);
// out[rows_b] = M1_b in1[rows_b] - M2_b in2[list2_b]  (+ fused prolongation / plain copy via pa);
// segment 1: the block's own rows, bs x bs matrices; segment 2: list + bs x kstride (32 | 64) matrices
bool block_apply2_ok(int bs, int k2);
void launch_block_apply2_b(hipStream_t st, const GroupTab& gt, int bs, int nblocks, const int* bptr,
                           const int* rows, const GroupPtrs& m1, const Seg2& s1, const GroupPtrs& m2,
                           const Seg2& s2, double* out, int ldo, size_t gso, int m, const ProlongArgs& pa);
void launch_block_apply2_b(hipStream_t st, const GroupTab& gt, int bs, int nblocks, const int* bptr,
                           const int* rows, const GroupPtrsF& m1, const Seg2& s1, const GroupPtrsF& m2,
                           const Seg2& s2, double* out, int ldo, size_t gso, int m, const ProlongArgs& pa);
// per shift:  out[b] = Ainv[b] * (alpha dE[b] + beta dA[b] + dJ[b])   (dense slices of S*Y, bs x ks)
void launch_ady_blocks(hipStream_t st, int nshift, int nblocks, int bs, int ks, const double* dA,
                       const double* dE, const double* dJ, const double* dT, const double* alphas,
                       const double* betas, const GroupPtrs& ainv, const GroupPtrs& out);
void launch_gt_blocks(hipStream_t st, int nshift, int nblocks, int bs, int ks, const double* jtd,
                      const GroupPtrs& ainv, const GroupPtrs& out);
void launch_to_f32(hipStream_t st, int nrows, int ncols, const double* src, int lds_, float* dst,
                   int ldd);
void launch_to_f32_tiled(hipStream_t st, int k, const double* src, float* dst);
void launch_gemm_tn_b(hipStream_t st, const GroupTab& gt, int n, int p, int q, const double* A,
                      int lda, const double* B, int ldb, size_t gsB, double* C, int ldc, size_t gsC);
// A given per group (A[g] is n x p, leading dimension lda)
void launch_gemm_nn_bp(hipStream_t st, const GroupTab& gt, int n, int p, int q, const GroupPtrs& A,
                       int lda, const double* C, int ldc, size_t gsC, double* Y, int ldy, size_t gsY,
                       double alpha, double beta);
void launch_gemm_nn_b(hipStream_t st, const GroupTab& gt, int n, int p, int q, const double* A,
                      int lda, const double* C, int ldc, size_t gsC, double* Y, int ldy, size_t gsY,
                      double alpha, double beta);
void launch_spmm(hipStream_t st, int nrows, const int* rp, const int* ci, const double* val,
                 const double* x, int ldx, const int* xmap, double* y, int ldy, const double* r,
                 int ldr, double alpha, double beta_r, const double* rowscale, int m);
size_t spmm_blocked_lds_bytes(int m, int max_cols, int max_nnz);
void launch_gather_vals(hipStream_t st, int nnz, const int* perm, const double* src, double* dst);
void launch_assemble_shift(hipStream_t st, int nnz, const double* srcA, const double* srcE,
                           const double* srcJ, double alpha, double beta, double* out);
void launch_axpby(hipStream_t st, size_t n, double a, const double* x, double b, double* y);
void launch_copy_cols(hipStream_t st, int nrows, int w, const double* src, int lds_, int sc0,
                      double* dst, int ldd, int dc0, double scale);
int dots_num_blocks(int nrows);
void launch_cols_dots(hipStream_t st, int nrows, int m, int nvec, const double* basis,
                      size_t vstride, const double* w, int want_self, double* partial,
                      double* out);
void launch_cols_update(hipStream_t st, int nrows, int m, int nvec, const double* basis,
                        size_t vstride, const double* h, double sign, const double* w,
                        const double* scale, double* out);
// setup kernels for up to RICADI_MAX_GROUPS shifts per launch (blockIdx.y = shift)
void launch_schur_blocks_bj(hipStream_t st, int nshift, int nblocks, int bs, const int* bptr,
                            const int* jd_ptr, const int* jd_vblk, const double* jd_val,
                            const GroupPtrs& bvinv, const GroupPtrs& blocks);
void launch_block_invert(hipStream_t st, int nshift, int nblocks, int bs, const int* bptr,
                         const GroupPtrs& blocks, int* flag);
void launch_block_combine(hipStream_t st, size_t n, const double* Ba, const double* Be,
                          double alpha, double beta, double* out);
void launch_gemm_tn(hipStream_t st, int n, int p, int q, const double* A, int lda, const double* B,
                    int ldb, double* C, int ldc);
void launch_gemm_nn(hipStream_t st, int n, int p, int q, const double* A, int lda, const double* C,
                    int ldc, double* Y, int ldy, double alpha, double beta);
int tsqr_num_blocks(int nrows);
void launch_tsqr_local(hipStream_t st, int nrows, int w, const double* A, int lda, double* Qloc,
                       double* Rstack);
void launch_tsqr_apply(hipStream_t st, int nrows, int w, const double* Qloc, const double* G,
                       double* Qout, int ldq);
void launch_cholqr_small(hipStream_t st, int w, const double* G, const double* Rprev, double* T,
                         double* R, int* flag);
void launch_cholqr_wide(hipStream_t st, int w, const double* G, int ldg, double* T, double* R, int* flag);
void launch_select_evecs(hipStream_t st, int c, int k, const double* evec, double* sel);
void launch_transpose_sign(hipStream_t st, int k, int k1, double sneg, const double* in, double* out);
// batched block Gauss-Jordan inverse of the coarse matrices (ricadi_kernels.hip); nb <= RICADI_MAX_GROUPS matrices
int gj_block();
int gj_max_batch();
void launch_gj_prep(hipStream_t st, int nb, double* const* mats, int k, int k0, int nbe, double* Cb, double* Rp,
                    double* D);
void launch_gj_diag(hipStream_t st, int nb, double* D, int nbe, int* flag);
void launch_gj_rows(hipStream_t st, int nb, double* const* mats, int k, int k0, int nbe, const double* Rb);
void launch_combine3(hipStream_t st, size_t n, const double* a0, const double* a1,
                     const double* a2, double alpha, double beta, double* out);

// dst (BF16 bit patterns, round to nearest even) = src (FP64), n entries
void launch_to_bf16(hipStream_t st, size_t n, const double* src, uint16_t* dst);
// The hot-shape sweeps (32-row blocks, 16 columns, fixed-stride records in pa.bmeta) on BF16-stored blocks: same
// contracts as launch_block_apply2_b / launch_block_apply_rect_b / launch_pressure_step_b with the FP32 panel; return
// false (nothing launched) when the shape is not the hot one.
bool launch_block_two32_h(hipStream_t st, const GroupTab& gt, int nblocks, const GroupPtrsH& m1, const Seg2& s1,
                          const GroupPtrsH& m2, const Seg2& s2, double* out, size_t gso, const ProlongArgs& pa,
                          bool f32_matrix_cores = false);
bool launch_block_rect32_h(hipStream_t st, const GroupTab& gt, int ks, int nblocks, const GroupPtrsH& mats,
                           const double* in, size_t gsi, double* out, size_t gso, int subtract, const ProlongArgs& pa);
void launch_pressure_step_h(hipStream_t st, const GroupTab& gt, int nblocks, const int* meta,
                            const GroupPtrsH& inv, const int* jci, const double* jv, bool with_sy, const int* syci,
                            const GroupPtrs& syv, const double* ec, size_t gse, const double* rp_, const _Float16* rp16,
                            size_t gsr, double* out, size_t gso, const ProlongArgs& pa, const float* zv32, size_t gsz32);

// K2p: the pressure step of the SIMPLE cycle fused into one launch (m = 16, 32 x 32 Schur blocks):
//   out[rows_b] = inv_b (J z + (S Y)_p ec - r_p)[rows_b]  with the epilogue options of the Schur sweep (pa).
// meta: per (block, row of the block) five ints {pressure-local row or -1, J row range [k0, k1), (S Y) pressure-row
// range [s0, s1)} at stride 5 (ricadi_ctx::ps_meta); with_sy = false: no coarse term; z is the n x 16 panel whose
// velocity rows are read, rp_ / rp16 the pressure rows of the residual (FP64 or FP16-stored), out the pressure rows of z.
// zv32 (optional, group stride gsz32): the velocity rows as the FP32 panel the first sweep left (64-B row gathers).
void launch_pressure_step_b(hipStream_t st, const GroupTab& gt, int nblocks, const int* meta,
                            const GroupPtrsF& inv, const int* jci, const double* jv, const double* z,
                            size_t gsz, bool with_sy, const int* syci, const GroupPtrs& syv, const double* ec, size_t gse,
                            const double* rp_, const _Float16* rp16, size_t gsr, double* out, size_t gso,
                            const ProlongArgs& pa, const float* zv32 = nullptr, size_t gsz32 = 0);
void launch_pressure_step_b(hipStream_t st, const GroupTab& gt, int nblocks, const int* meta,
                            const GroupPtrs& inv, const int* jci, const double* jv, const double* z,
                            size_t gsz, bool with_sy, const int* syci, const GroupPtrs& syv, const double* ec, size_t gse,
                            const double* rp_, const _Float16* rp16, size_t gsr, double* out, size_t gso,
                            const ProlongArgs& pa, const float* zv32 = nullptr, size_t gsz32 = 0);

// K3h: last Arnoldi pass + Hessenberg / Givens update in one launch (FP16 basis, m = 16)
bool update_hess_fused_ok(int m, bool fp16_basis);
void launch_cols_update16_hess_b(hipStream_t st, const GroupTab& gt, int nrows, int nvec, const _Float16* basis,
                                 size_t vstride, size_t gsb, const double* h1, const double* h2, size_t gsh, int use_sum,
                                 const double* w, size_t gsw, double* out, size_t gso, _Float16* outf, size_t gsf, int j,
                                 int restart, double* H, double* cs, double* sn, double* g, const double* resid_in,
                                 double* resid_out, const double* bnorm, double tol, double* host_resid,
                                 const float* w32 = nullptr);

// First two Arnoldi passes of a 16-column panel against the FP16-stored basis with the panel w stored in FP32 (the
// operator's output on the hot path, round 4); the second pass leaves w as it is ("w kept" form).  launch_cols_update16_
// hess_b takes the same panel through its w32 argument.
bool arnoldi16_w32_ok(int nvec_max);
void launch_cols_dots16_w32(hipStream_t st, const GroupTab& gt, int nrows, int nvec, const _Float16* basis,
                            size_t vstride, size_t gsb, const float* w32, size_t gsw, double* partial, size_t gsp,
                            double* out, size_t gso);
void launch_cols_update_dots16_w32(hipStream_t st, const GroupTab& gt, int nrows, int nvec, const _Float16* basis,
                                   size_t vstride, size_t gsb, const double* h, size_t gsh, float* w32, size_t gsw,
                                   double* partial, size_t gsp, double* out, size_t gso);

// K5c: pivoted Cholesky of a (possibly augmented) symmetric matrix, 8 / 16 / 32 pivots per launch pair --
// the eigensolver-free recompression (ricadi_kernels.hip).  State lives on the device so that the host can
// issue all blocks without a read-back: once `stop` is set the remaining launches return at once.
struct PcholState {
  double d0;     // first pivot (scale of the tolerance)
  int rank;      // rows of the factor written so far
  int stop;      // 1: tolerance / row limit reached
  int nblk;      // rows written by the last panel launch
  int pad;
};
int pchol_block(int nc);          // pivots per panel launch for nc columns (0: nc too large)
void launch_pchol_panel(hipStream_t st, const double* A, int ld, int nr, int nc, double tol, int kmax,
                        PcholState* stt, double* Rout, int ldr, int* done);
void launch_pchol_trail(hipStream_t st, double* A, int ld, int nr, int nc, const PcholState* stt,
                        const double* Rall, int ldr);
void launch_transpose(hipStream_t st, int rows, int cols, const double* in, int ldi, double* out, int ldo);

// K4s: all Z blocks of an ADI sweep + their squared column norms in two launches
bool sweep_combine_ok(int m, int nslot, int G);
size_t sweep_combine_partial_len(int nrows, int m, int G);
void launch_sweep_combine(hipStream_t st, int nrows, int m, int nslot, int G, const double* U, size_t ustride,
                          const double* coef, double* Z, int zld, int zc0, double* partial, double* norms2);

void set_error(const std::string& msg);

}  // namespace ricadi
