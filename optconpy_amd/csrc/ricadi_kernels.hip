// ricadi_kernels.hip -- CDNA4 (gfx950) kernels of the Newton-ADI hot path.
//
// Everything here is new code: the reference (/root/reference) has no native or
// GPU source at all; the kernels implement the list
// K1..K6 of SURVEY.md section 8(a).
//
// Layout rules shared by all kernels
//   * dense panels are row-major n x m, one row = m contiguous doubles
//     (m = 16 -> one 128-B line per row: an indexed row gather is a full line);
//   * a wavefront (64 lanes) is split into 16-lane groups; a group owns one
//     matrix row and its lanes own the panel columns g, g+16, ...;
//   * reductions over rows are two-stage (per-workgroup partials, then a small
//     reduce kernel), so results are bitwise reproducible run to run.
#include <cstdlib>
#include <type_traits>

#include "ricadi_internal.h"

namespace ricadi {

typedef double d4 __attribute__((ext_vector_type(4)));

// Broadcast lane T of every 16-lane row to the whole row on the VALU
// (DPP row_newbcast, gfx90a+): no LDS instruction, unlike __shfl/ds_bpermute.
// The SpMM kernels were LDS-pipe bound by their broadcasts (SQ_ACTIVE_INST_LDS
// ~72 % of the kernel, profiles/r01_spmm_pmc.txt).
template <int T>
__device__ __forceinline__ int bc16i(int v) {
  // mov_dpp: "old" operand undefined + bound_ctrl, so no zero-initialising v_mov
  return __builtin_amdgcn_mov_dpp(v, 0x150 + T, 0xF, 0xF, true);
}
template <int T>
__device__ __forceinline__ double bc16d(double v) {
  // one v_mov_b64_dpp (row_newbcast is the one DPP control 64-bit moves accept)
  return __builtin_amdgcn_update_dpp(v, v, 0x150 + T, 0xF, 0xF, true);
}
#define RICADI_FOR16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define RICADI_FOR8A(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define RICADI_FOR8B(M) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

// ---------------------------------------------------------------------------
// K1: CSR SpMM on row-major panels.
//   y[i,:] = beta_r * r[i,:] + alpha * rowscale[i] * sum_k val[k] * x[xrow(col[k]),:]
// xmap (optional) redirects the gathered row (used to apply S to a prolongated
// coarse vector without materialising it).  One 16-lane group per row; the
// (col,val) loads are group-uniform (one request), the x-row load is one
// coalesced 128-B line per 16 columns.
// ---------------------------------------------------------------------------
// Low-rank epilogue shared by the SpMM kernels:  (U (V^T x))[row, col] with the
// q x m coefficients V^T x already reduced (lrc).  q is small (the number of inputs).
__device__ __forceinline__ double lowrank_term(const LowRankArgs& lr, const double* __restrict__ lrc,
                                               int row, int col, int m) {
  const double* __restrict__ u = lr.U + (size_t)row * lr.q;
  double s = 0.0;
  for (int k = 0; k < lr.q; ++k) s = fma(u[k], lrc[k * m + col], s);
  return s;
}

template <int CPL>
__global__ __launch_bounds__(256) void spmm_kernel(
    GroupTab gt, int nrows, const int* __restrict__ rp, const int* __restrict__ ci,
    GroupPtrs vals, const double* __restrict__ x, int ldx, size_t gsx,
    const int* __restrict__ xmap, double* __restrict__ y, int ldy, size_t gsy,
    const double* __restrict__ r, int ldr, size_t gsr, double alpha, double beta_r,
    const double* __restrict__ rowscale, int m, LowRankArgs lr) {
  const int grp = gt.gid[blockIdx.z];
  const double* __restrict__ val = vals.p[grp];
  x += (size_t)grp * gsx;
  y += (size_t)grp * gsy;
  if (r) r += (size_t)grp * gsr;
  const double* __restrict__ lrc = lr.c + (size_t)grp * lr.gsc;
  const int g = threadIdx.x & 15;
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (row >= nrows) return;
  double acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = 0.0;
  const int k0 = rp[row], k1 = rp[row + 1];
  int k = k0;
  for (; k + 1 < k1; k += 2) {
    int c0 = ci[k], c1 = ci[k + 1];
    const double v0 = val[k], v1 = val[k + 1];
    if (xmap) { c0 = xmap[c0]; c1 = xmap[c1]; }
    const double* x0 = x + (size_t)c0 * ldx;
    const double* x1 = x + (size_t)c1 * ldx;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int col = g + 16 * c;
      if (col < m) {
        acc[c] = fma(v0, x0[col], acc[c]);
        acc[c] = fma(v1, x1[col], acc[c]);
      }
    }
  }
  if (k < k1) {
    int c0 = ci[k];
    const double v0 = val[k];
    if (xmap) c0 = xmap[c0];
    const double* x0 = x + (size_t)c0 * ldx;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int col = g + 16 * c;
      if (col < m) acc[c] = fma(v0, x0[col], acc[c]);
    }
  }
  const double sc = alpha * (rowscale ? rowscale[row] : 1.0);
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int col = g + 16 * c;
    if (col < m) {
      double out = sc * acc[c];
      if (r) out += beta_r * r[(size_t)row * ldr + col];
      if (row < lr.nrows) out -= lowrank_term(lr, lrc, row, col, m);
      y[(size_t)row * ldy + col] = out;
    }
  }
}

// Variant 2 (default).  The 16 lanes of a row group load 16 consecutive
// (col, val) pairs with ONE coalesced load each and broadcast them with
// width-16 shuffles, so the 16 x-row gathers of a chunk are independent and all
// in flight together (the variant above serialises col -> gather per entry).
// Padding entries use val = 0 / col = 0, i.e. a harmless cached gather,
// so the inner loop is branch free.  Row blocks are dealt to the 8 XCDs in
// contiguous ranges (blockIdx % 8 selects the range), which keeps the gathered
// x rows of a band matrix inside that XCD's L2.
// CHK = entries per chunk (16, or 8 for matrices with short rows such as S*Y: half the
// broadcast steps are saved when a row has <= 8 entries).
template <int CPL, int CHK, class XT = double, class RT = double>
__global__ __launch_bounds__(256) void spmm_kernel_v2(
    GroupTab gt, int nrows, const int* __restrict__ rp, const int* __restrict__ ci,
    GroupPtrs vals, const XT* __restrict__ x, int ldx, size_t gsx,
    const int* __restrict__ xmap, double* __restrict__ y, int ldy, size_t gsy,
    const RT* __restrict__ r, int ldr, size_t gsr, double alpha, double beta_r,
    const double* __restrict__ rowscale, int m, LowRankArgs lr) {
  const int grp = gt.gid[blockIdx.z];
  const double* __restrict__ val = vals.p[grp];
  x += (size_t)grp * gsx;
  y += (size_t)grp * gsy;
  if (r) r += (size_t)grp * gsr;
  const double* __restrict__ lrc = lr.c + (size_t)grp * lr.gsc;
  // bijective XCD remap of the block index (cdna guide, T1)
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int qd = nwg >> 3, rm = nwg & 7, xcd = orig & 7;
  const int blk = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (orig >> 3);
  const int g = threadIdx.x & 15;
  const int row = blk * 16 + (threadIdx.x >> 4);
  const bool live = row < nrows;
  double acc[CPL];
  int colx[CPL];      // lanes beyond m read column 0 (branch-free loop); their result is dropped
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    acc[c] = 0.0;
    colx[c] = (g + 16 * c < m) ? g + 16 * c : 0;
  }
  const int k0 = live ? rp[row] : 0, k1 = live ? rp[row + 1] : 0;
  // all 4 groups of the wave iterate the same number of chunks (shuffles need
  // every lane): take the wave-wide maximum
  int nch = (k1 - k0 + CHK - 1) / CHK;
  nch = max(nch, __shfl_xor(nch, 16, 64));
  nch = max(nch, __shfl_xor(nch, 32, 64));
  for (int ch = 0; ch < nch; ++ch) {
    const int k = k0 + ch * CHK + g;
    int myc = 0;            // padding: val = 0 times row 0 of x (always a valid row;
    double myv = 0.0;       // the matrix may be rectangular, so "own row" is not)
    if (g < CHK && k < k1) {
      myc = ci[k];
      myv = val[k];
      if (xmap) myc = xmap[myc];
    }
#define RICADI_V2_STEP(T)                                             \
  {                                                                   \
    const int c0 = bc16i<T>(myc);                                     \
    const double v0 = bc16d<T>(myv);                                  \
    const XT* x0 = x + (size_t)c0 * ldx;                              \
    _Pragma("unroll") for (int c = 0; c < CPL; ++c)                   \
        acc[c] = fma(v0, (double)x0[colx[c]], acc[c]);                \
  }
    if (CHK == 16) {
      RICADI_FOR16(RICADI_V2_STEP)
    } else {
      RICADI_V2_STEP(0) RICADI_V2_STEP(1) RICADI_V2_STEP(2) RICADI_V2_STEP(3)
      RICADI_V2_STEP(4) RICADI_V2_STEP(5) RICADI_V2_STEP(6) RICADI_V2_STEP(7)
    }
#undef RICADI_V2_STEP
  }
  if (!live) return;
  const double sc = alpha * (rowscale ? rowscale[row] : 1.0);
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int col = g + 16 * c;
    if (col < m) {
      double out = sc * acc[c];
      if (r) out += beta_r * (double)r[(size_t)row * ldr + col];
      if (row < lr.nrows) out -= lowrank_term(lr, lrc, row, col, m);
      y[(size_t)row * ldy + col] = out;
    }
  }
}

static int spmm_variant() { return 2; }

static void spmm_dispatch(hipStream_t st, const GroupTab& gt, int nrows, const int* rp,
                          const int* ci, const GroupPtrs& vals, const double* x, int ldx,
                          size_t gsx, const int* xmap, double* y, int ldy, size_t gsy,
                          const double* r, int ldr, size_t gsr, double alpha, double beta_r,
                          const double* rowscale, int m, const LowRankArgs& lr = LowRankArgs(),
                          int chunk = 16) {
  if (nrows <= 0 || m <= 0 || gt.ng <= 0) return;
  dim3 grid((nrows + 15) / 16, 1, gt.ng), block(256);
  const int cpl = (m + 15) / 16;
  const bool v2 = spmm_variant() == 2;
  if (v2 && chunk == 8 && cpl <= 2) {
    if (cpl == 1)
      hipLaunchKernelGGL((spmm_kernel_v2<1, 8>), grid, block, 0, st, gt, nrows, rp, ci, vals, x, ldx,
                         gsx, xmap, y, ldy, gsy, r, ldr, gsr, alpha, beta_r, rowscale, m, lr);
    else
      hipLaunchKernelGGL((spmm_kernel_v2<2, 8>), grid, block, 0, st, gt, nrows, rp, ci, vals, x, ldx,
                         gsx, xmap, y, ldy, gsy, r, ldr, gsr, alpha, beta_r, rowscale, m, lr);
    return;
  }
#define RICADI_SPMM_CASE(C)                                                              \
  case C:                                                                                \
    if (v2)                                                                              \
      hipLaunchKernelGGL((spmm_kernel_v2<C, 16>), grid, block, 0, st, gt, nrows, rp, ci, vals, \
                         x, ldx, gsx, xmap, y, ldy, gsy, r, ldr, gsr, alpha, beta_r,     \
                         rowscale, m, lr);                                               \
    else                                                                                 \
      hipLaunchKernelGGL(spmm_kernel<C>, grid, block, 0, st, gt, nrows, rp, ci, vals, x, \
                         ldx, gsx, xmap, y, ldy, gsy, r, ldr, gsr, alpha, beta_r,        \
                         rowscale, m, lr);                                               \
    break;
  switch (cpl) {
    RICADI_SPMM_CASE(1)
    RICADI_SPMM_CASE(2)
    RICADI_SPMM_CASE(3)
    RICADI_SPMM_CASE(4)
    RICADI_SPMM_CASE(5)
    RICADI_SPMM_CASE(6)
    RICADI_SPMM_CASE(7)
    RICADI_SPMM_CASE(8)
    default:
      break;  // m <= RICADI_MAX_M = 128 is enforced by the callers
  }
#undef RICADI_SPMM_CASE
}
void launch_spmm(hipStream_t st, int nrows, const int* rp, const int* ci, const double* val,
                 const double* x, int ldx, const int* xmap, double* y, int ldy,
                 const double* r, int ldr, double alpha, double beta_r,
                 const double* rowscale, int m) {
  spmm_dispatch(st, single_group(), nrows, rp, ci, same_ptr(val), x, ldx, 0, xmap, y, ldy, 0, r, ldr,
                0, alpha, beta_r, rowscale, m);
}
void launch_spmm_b(hipStream_t st, const GroupTab& gt, int nrows, const int* rp, const int* ci,
                   const GroupPtrs& vals, const double* x, int ldx, size_t gsx, const int* xmap,
                   double* y, int ldy, size_t gsy, const double* r, int ldr, size_t gsr,
                   double alpha, double beta_r, int m, const LowRankArgs& lr, int chunk) {
  spmm_dispatch(st, gt, nrows, rp, ci, vals, x, ldx, gsx, xmap, y, ldy, gsy, r, ldr, gsr, alpha,
                beta_r, nullptr, m, lr, chunk);
}

// Forms with an operand taken from the FP16-stored Krylov vector (panels of <= 16 columns): x16 replaces x
// (restriction of the current vector), r16 replaces r (its pressure rows as the additive term)
void launch_spmm_h(hipStream_t st, const GroupTab& gt, int nrows, const int* rp, const int* ci,
                   const GroupPtrs& vals, const double* x, const _Float16* x16, int ldx, size_t gsx, double* y, int ldy,
                   size_t gsy, const _Float16* r16, int ldr, size_t gsr, double alpha, double beta_r, int m, int chunk) {
  if (nrows <= 0 || m <= 0 || m > 16 || gt.ng <= 0) return;
  dim3 grid((nrows + 15) / 16, 1, gt.ng), block(256);
  const int* nomap = nullptr;
  const double* norow = nullptr;
  if (x16 && chunk == 8)
    hipLaunchKernelGGL((spmm_kernel_v2<1, 8, _Float16, _Float16>), grid, block, 0, st, gt, nrows, rp, ci, vals, x16,
                       ldx, gsx, nomap, y, ldy, gsy, r16, ldr, gsr, alpha, beta_r, norow, m, LowRankArgs());
  else if (x16)
    hipLaunchKernelGGL((spmm_kernel_v2<1, 16, _Float16, _Float16>), grid, block, 0, st, gt, nrows, rp, ci, vals, x16,
                       ldx, gsx, nomap, y, ldy, gsy, r16, ldr, gsr, alpha, beta_r, norow, m, LowRankArgs());
  else if (chunk == 8)
    hipLaunchKernelGGL((spmm_kernel_v2<1, 8, double, _Float16>), grid, block, 0, st, gt, nrows, rp, ci, vals, x, ldx,
                       gsx, nomap, y, ldy, gsy, r16, ldr, gsr, alpha, beta_r, norow, m, LowRankArgs());
  else
    hipLaunchKernelGGL((spmm_kernel_v2<1, 16, double, _Float16>), grid, block, 0, st, gt, nrows, rp, ci, vals, x, ldx,
                       gsx, nomap, y, ldy, gsy, r16, ldr, gsr, alpha, beta_r, norow, m, LowRankArgs());
}

// ---------------------------------------------------------------------------
// K1, LDS-tiled variant for the saddle operator.
//
// Rows are processed in blocks of <= 64 rows that form a compact patch of the
// mesh (pairs of block-Jacobi aggregates), listed in `rows` -- the panels keep
// the caller's row order: with m = 16 a panel row is one 128-B line, so neither
// the gather of x rows nor the scatter of y rows needs neighbouring rows to be
// neighbours in memory.  Per block:
//   phase 1  the block's DISTINCT x rows (cols[cptr[b]..)) are loaded once into
//            an LDS tile (one coalesced 128-B row per 16-lane group and load,
//            all loads of a thread independent), and the block's slice of the
//            matrix (values + 16-bit local column indices, contiguous in block
//            order) is streamed into LDS with fully coalesced loads;
//   phase 2  every 16-lane group accumulates its rows from LDS only.
// A row of x is thus read from L2/HBM once per block instead of once per
// non-zero (the v2 kernel re-gathers every row ~28 times through the vector L1).
// ---------------------------------------------------------------------------
// Block metadata comes PADDED to fixed strides -- rows2[b][32] (global row, -1 =
// none), rp2[b][33] (entry ranges in block order), cols2[b][max_cols] (gathered x
// row per tile slot, -1 = none; for the coarse-residual launch the aggregate map is
// already applied) -- so every address of the first round of loads follows from
// the block index alone: the kernel is bound by the latency of its dependent
// loads, and this removes one full round trip (block pointers -> row/column lists).
// HAS_R / HAS_LR: compile the residual term / the low-rank epilogue in (the plain
// operator launch of the GMRES iteration has neither).
template <bool HAS_R, bool HAS_LR, class XT = double>
__global__ __launch_bounds__(256) void spmm_blocked_kernel(
    const int* __restrict__ rows2, const int* __restrict__ rp2, const int* __restrict__ cols2,
    const uint16_t* __restrict__ lidx, GroupTab gt, GroupPtrs vals,
    const XT* __restrict__ x, int ldx, size_t gsx,
    double* __restrict__ y, int ldy, size_t gsy, const double* __restrict__ r, int ldr,
    size_t gsr, double alpha, double beta_r, int m, int max_cols, LowRankArgs lr) {
  extern __shared__ double xs[];                             // max_cols x m
  // Groups ride in grid.z (group-major dispatch: consecutive workgroups are
  // neighbouring row blocks of ONE panel, whose gathered x rows overlap -- walking
  // the groups fastest instead, to share the matrix slice in L2, measured 13 %
  // slower, and building the values from shared (beta*A + J, E) arrays another 10 %:
  // the kernel is bound by the latency of its dependent gathers, not by HBM bytes).
  const int grp = gt.gid[blockIdx.z];
  const double* __restrict__ val = vals.p[grp];
  x += (size_t)grp * gsx;
  y += (size_t)grp * gsy;
  if (HAS_R) r += (size_t)grp * gsr;
  const double* __restrict__ lrc = HAS_LR ? lr.c + (size_t)grp * lr.gsc : nullptr;
  // XCD-contiguous block ranges (bijective remap, cdna guide T1)
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int qd = nwg >> 3, rm = nwg & 7, xcd = orig & 7;
  const int b = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (orig >> 3);
  const int g = threadIdx.x & 15, gq = threadIdx.x >> 4;
  const int* __restrict__ bcols = cols2 + (size_t)b * max_cols;
  // phase 0: the (value, local index) pairs of this group's two rows (blocks
  // hold <= 32 rows) are requested FIRST, 16 per lane-row and chunk, so that
  // they are in flight together with the x-tile gathers of phase 1.
  constexpr int NR = 2, NCH = 3;               // rows per group, 16-entry chunks held in registers
  int ka[NR], kb[NR], grow[NR];
  double myv[NR][NCH];
  int myl[NR][NCH];
#pragma unroll
  for (int rr = 0; rr < NR; ++rr) {
    const int q = gq + 16 * rr;
    ka[rr] = rp2[b * 33 + q];
    kb[rr] = rp2[b * 33 + q + 1];
    grow[rr] = rows2[b * 32 + q];
  }
#pragma unroll
  for (int rr = 0; rr < NR; ++rr)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int k = ka[rr] + ch * 16 + g;
      const bool ok = k < kb[rr];
      myv[rr][ch] = ok ? val[k] : 0.0;
      myl[rr][ch] = ok ? (int)lidx[k] : 0;   // NOT touched before the barrier: see below
    }
  // phase 1: x tile.  Indices first, then ALL gathers of the thread, then the
  // LDS stores -- so that the loads are in flight together (a load followed by
  // its own ds_write makes hipcc wait vmcnt(0) per row).
  constexpr int XJ = 5;                        // 16 groups x 5 = 80 tile rows per pass
  for (int cc = g; cc < m; cc += 16) {
    for (int jb = 0; jb < max_cols; jb += 16 * XJ) {
      int cidx[XJ];
      XT xv[XJ];             // raw loads; an FP32 x is converted at the LDS store, not between the loads
#pragma unroll
      for (int t = 0; t < XJ; ++t) {
        const int j = jb + gq + 16 * t;
        cidx[t] = (j < max_cols) ? bcols[j] : -1;
      }
#pragma unroll
      for (int t = 0; t < XJ; ++t) xv[t] = (cidx[t] >= 0) ? x[(size_t)cidx[t] * ldx + cc] : (XT)0;
#pragma unroll
      for (int t = 0; t < XJ; ++t) {
        const int j = jb + gq + 16 * t;
        if (cidx[t] >= 0) xs[j * m + cc] = (double)xv[t];
      }
    }
  }
  __syncthreads();
  // Local index -> BYTE offset of the tile row, once per entry.  Done here and
  // not at load time: using a phase-0 value before the barrier makes the wave
  // wait for those loads before it has issued the x-tile gathers (measured:
  // 198 us instead of 168 us at n = 5e5).
  const unsigned rowbytes = (unsigned)m * 8u;
#pragma unroll
  for (int rr = 0; rr < NR; ++rr)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) myl[rr][ch] *= (int)rowbytes;
  // The DPP operands below are read by hand-written DPP instructions: keep the
  // VALU writes above two wait states away from them (hipcc pads nothing for asm).
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 2");
  __builtin_amdgcn_sched_barrier(0);
  // phase 2.  Per (row, entry) step and wave: ONE v_add_u32_dpp (row_newbcast of
  // the entry's tile-row offset + this lane's column address), ONE ds_read_b64
  // and ONE v_fmac_f64_dpp (row_newbcast of the value fused into the FP64 FMA;
  // gfx90a+ allows row_newbcast on 64-bit DPP ALU ops).  Compiler-generated
  // code for the same step was 9 VALU instructions (profiles/r01_spmm_pmc.txt).
  typedef __attribute__((address_space(3))) const double lds_cdouble;
  const unsigned xs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) double*)xs;
#define RICADI_TILE_STEP(T)                                                                  \
  {                                                                                          \
    unsigned ad;                                                                             \
    asm("v_add_u32_dpp %0, %1, %2 row_newbcast:" #T " row_mask:0xf bank_mask:0xf"            \
        : "=v"(ad)                                                                           \
        : "v"(lcur), "v"(lane_base));                                                        \
    const double xv = *(lds_cdouble*)(size_t)ad;                                             \
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #T " row_mask:0xf bank_mask:0xf"           \
        : "+v"(acc[(T)&3])                                                                   \
        : "v"(vcur), "v"(xv));                                                               \
  }
#pragma unroll
  for (int rr = 0; rr < NR; ++rr) {
    const bool live = grow[rr] >= 0;
    // HALF chunks (8 entries) needed by any of the wave's four groups (DPP needs all lanes): 72 % of the velocity
    // rows of a Taylor-Hood operator hold 17-24 entries, most pressure rows 33-40 -- with whole 16-entry chunks a
    // fifth of the steps were padding (mean 35.4 steps per row against 27.3 entries; 29.5 with half chunks)
    int nh = (kb[rr] - ka[rr] + 7) >> 3;
    nh = max(nh, __shfl_xor(nh, 16, 64));
    nh = max(nh, __shfl_xor(nh, 32, 64));
    const int nch = (nh + 1) >> 1;
    for (int cc = g; cc < m + (16 - (m & 15)) % 16; cc += 16) {
      const int ccs = cc < m ? cc : 0;         // lanes beyond m stay in the broadcasts
      const unsigned lane_base = xs_lds + (unsigned)ccs * 8u;
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        if (2 * ch < nh) {
          const int lcur = myl[rr][ch];
          const double vcur = myv[rr][ch];
          RICADI_FOR8A(RICADI_TILE_STEP)
          if (2 * ch + 1 < nh) { RICADI_FOR8B(RICADI_TILE_STEP) }
        }
      }
      // rows longer than NCH*16 entries: stream the rest
      for (int ch = NCH; ch < nch; ++ch) {
        const int k = ka[rr] + ch * 16 + g;
        int lcur = 0;
        double vcur = 0.0;
        if (k < kb[rr]) {
          lcur = (int)lidx[k] * (int)rowbytes;
          vcur = val[k];
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 2");
        __builtin_amdgcn_sched_barrier(0);
        RICADI_FOR8A(RICADI_TILE_STEP)
        if (2 * ch + 1 < nh) { RICADI_FOR8B(RICADI_TILE_STEP) }
      }
      if (live && cc < m) {
        const int row = grow[rr];
        double out = alpha * ((acc[0] + acc[1]) + (acc[2] + acc[3]));
        if (HAS_R) out += beta_r * r[(size_t)row * ldr + cc];
        if (HAS_LR && row < lr.nrows) out -= lowrank_term(lr, lrc, row, cc, m);
        y[(size_t)row * ldy + cc] = out;
      }
    }
  }
#undef RICADI_TILE_STEP
}
size_t spmm_blocked_lds_bytes(int m, int max_cols, int max_nnz) {
  (void)max_nnz;
  return (size_t)max_cols * m * sizeof(double) + 16;
}
void launch_spmm_blocked_b(hipStream_t st, const GroupTab& gt, int nblk, const int* rows2,
                           const int* rp2, const int* cols2, const uint16_t* lidx,
                           const GroupPtrs& vals, const double* x, int ldx, size_t gsx, double* y,
                           int ldy, size_t gsy, const double* r, int ldr, size_t gsr, double alpha,
                           double beta_r, int m, int max_cols, const LowRankArgs& lr) {
  if (nblk <= 0 || gt.ng <= 0) return;
  const dim3 grid(nblk, 1, gt.ng), block(256);
  const size_t lds = spmm_blocked_lds_bytes(m, max_cols, 0);
#define RICADI_TILE_LAUNCH(R, L)                                                                  \
  hipLaunchKernelGGL((spmm_blocked_kernel<R, L>), grid, block, lds, st, rows2, rp2, cols2, lidx, gt, \
                     vals, x, ldx, gsx, y, ldy, gsy, r, ldr, gsr, alpha, beta_r, m, max_cols, lr)
  const bool has_lr = lr.q > 0 && lr.nrows > 0;
  if (r && has_lr) RICADI_TILE_LAUNCH(true, true);
  else if (r) RICADI_TILE_LAUNCH(true, false);
  else if (has_lr) RICADI_TILE_LAUNCH(false, true);
  else RICADI_TILE_LAUNCH(false, false);
#undef RICADI_TILE_LAUNCH
}

// plain operator product with an FP32-stored x (the flexible GMRES applies S to the stored Z_j): the x tile
// is converted while it is staged, the inner loop is the same
void launch_spmm_blocked_x32(hipStream_t st, const GroupTab& gt, int nblk, const int* rows2, const int* rp2,
                             const int* cols2, const uint16_t* lidx, const GroupPtrs& vals, const float* x, int ldx,
                             size_t gsx, double* y, int ldy, size_t gsy, double alpha, int m, int max_cols) {
  if (nblk <= 0 || gt.ng <= 0) return;
  const dim3 grid(nblk, 1, gt.ng), block(256);
  const size_t lds = spmm_blocked_lds_bytes(m, max_cols, 0);
  hipLaunchKernelGGL((spmm_blocked_kernel<false, false, float>), grid, block, lds, st, rows2, rp2, cols2, lidx, gt,
                     vals, x, ldx, gsx, y, ldy, gsy, (const double*)nullptr, 0, (size_t)0, alpha, 0.0, m, max_cols,
                     LowRankArgs());
}

// ---------------------------------------------------------------------------
// K1, multi-shift form of the LDS-tiled kernel (the "batched shifted" kernel of SURVEY.md
// App. C.4 / section 8d): the shifted matrices of a sweep differ by two scalars only,
//     S(alpha_g, beta_g) = alpha_g * E + beta_g * A + J      on one sparsity pattern,
// so ONE workgroup serves a row block for ALL active groups: the block's slice of the three
// value arrays (block order) and its 16-bit local indices are loaded into registers once,
// the tile's column list once, and the groups are then walked in a software pipeline --
// while group g is accumulated out of LDS tile (g & 1), the x rows of group g+1 are already
// in flight into registers and go to the other tile behind the barrier.  Per launch the
// matrix is read once instead of once per group (26 B per non-zero instead of 10 B x G),
// and only the first group of a workgroup pays the dependent-load latency of the metadata.
// Panels up to 16 columns (one column per lane of a 16-lane row group).
// grid.y splits the active groups (blockIdx.y, blockIdx.y + gridDim.y, ...) when there are
// too few row blocks to fill the chip.
// ---------------------------------------------------------------------------
struct GroupCoefs {
  double alpha[RICADI_MAX_GROUPS], beta[RICADI_MAX_GROUPS];
};

// Value sources: vE (cal E part) and vAJ = (cal A part) + (J / J^T part) -- the two have
// disjoint supports (velocity-velocity entries vs. constraint entries), so
//     value = alpha_g * vE + (entry in the velocity-velocity block ? beta_g : 1) * vAJ,
// the block membership riding in bit 15 of the 16-bit local column index (tiles have at
// most 160 columns).
template <bool HAS_R, class XT = double>
__global__ __launch_bounds__(256) void spmm_blocked_ms_kernel(
    const int* __restrict__ rows2, const int* __restrict__ rp2, const int* __restrict__ cols2,
    const uint16_t* __restrict__ lidx, GroupTab gt, GroupCoefs cf,
    const double* __restrict__ vAJ, const double* __restrict__ vE,
    const XT* __restrict__ x, int ldx, size_t gsx, double* __restrict__ y, int ldy, size_t gsy,
    const double* __restrict__ r, int ldr, size_t gsr, double alpha, double beta_r, int m,
    int max_cols) {
  extern __shared__ double xs[];                             // 2 tiles of max_cols x 16
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int qd = nwg >> 3, rm = nwg & 7, xcd = orig & 7;
  const int b = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (orig >> 3);
  const int g = threadIdx.x & 15, gq = threadIdx.x >> 4;
  const int* __restrict__ bcols = cols2 + (size_t)b * max_cols;
  constexpr int NR = 2, NCH = 3;               // rows per 16-lane group, 16-entry chunks in registers
  constexpr int XJ = 5, XP = 2;                // tile rows per thread: XP passes of XJ (16 * 10 = 160 slots)
  int ka[NR], kb[NR], grow[NR];
  double mAJ[NR][NCH], mE[NR][NCH];
  int myl[NR][NCH];                            // byte offset in the tile | velocity-velocity flag (bit 30)
#pragma unroll
  for (int rr = 0; rr < NR; ++rr) {
    const int q = gq + 16 * rr;
    ka[rr] = rp2[b * 33 + q];
    kb[rr] = rp2[b * 33 + q + 1];
    grow[rr] = rows2[b * 32 + q];
  }
  const int gc = g < m ? g : 0;                // lanes beyond m shadow column 0 (kept in the broadcasts)
  // tile slots of this thread (the same for every group): byte offsets into a panel, -1 = none
  int xoff[XP][XJ];
#pragma unroll
  for (int pp = 0; pp < XP; ++pp)
#pragma unroll
    for (int t = 0; t < XJ; ++t) {
      const int j = pp * 16 * XJ + gq + 16 * t;
      const int ci = (j < max_cols) ? bcols[j] : -1;
      xoff[pp][t] = ci >= 0 ? (ci * ldx + gc) * (int)sizeof(XT) : -1;
    }
#pragma unroll
  for (int rr = 0; rr < NR; ++rr)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int k = ka[rr] + ch * 16 + g;
      const bool ok = k < kb[rr];
      mAJ[rr][ch] = ok ? vAJ[k] : 0.0;
      mE[rr][ch] = ok ? vE[k] : 0.0;
      myl[rr][ch] = ok ? (int)lidx[k] : 0;
    }
  const int ystep = gridDim.y;
  int gi = blockIdx.y;
  if (gi >= gt.ng) return;
  XT xv[XP][XJ];             // raw loads (converted when they go to the LDS tile)
  auto fetch = [&](int grp) {
    const char* __restrict__ xg = reinterpret_cast<const char*>(x + (size_t)grp * gsx);
#pragma unroll
    for (int pp = 0; pp < XP; ++pp)
#pragma unroll
      for (int t = 0; t < XJ; ++t)
        xv[pp][t] = (xoff[pp][t] >= 0) ? *reinterpret_cast<const XT*>(xg + (unsigned)xoff[pp][t]) : (XT)0;
  };
  auto stash = [&](int buf) {
    double* __restrict__ tile = xs + (size_t)buf * max_cols * 16;
#pragma unroll
    for (int pp = 0; pp < XP; ++pp)
#pragma unroll
      for (int t = 0; t < XJ; ++t) {
        const int j = pp * 16 * XJ + gq + 16 * t;
        if (j < max_cols) tile[j * 16 + g] = (double)xv[pp][t];
      }
  };
  fetch(gt.gid[gi]);
  stash(0);
  // local index -> byte offset within a tile (row = 16 doubles); flag moves to bit 30
#pragma unroll
  for (int rr = 0; rr < NR; ++rr)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int v = myl[rr][ch];
      myl[rr][ch] = ((v & 0x7fff) << 7) | ((v & 0x8000) << 15);
    }
  __syncthreads();
  typedef __attribute__((address_space(3))) const double lds_cdouble;
  const unsigned xs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) double*)xs;
  int nhr[NR];                                   // half chunks (8 entries) needed by any of the wave's four groups
#pragma unroll
  for (int rr = 0; rr < NR; ++rr) {
    int nh = (kb[rr] - ka[rr] + 7) >> 3;
    nh = max(nh, __shfl_xor(nh, 16, 64));
    nh = max(nh, __shfl_xor(nh, 32, 64));
    nhr[rr] = nh;
  }
  int buf = 0;
#define RICADI_MS_STEP(T)                                                                    \
  {                                                                                          \
    unsigned ad;                                                                             \
    asm("v_add_u32_dpp %0, %1, %2 row_newbcast:" #T " row_mask:0xf bank_mask:0xf"            \
        : "=v"(ad)                                                                           \
        : "v"(lcur), "v"(lane_base));                                                        \
    const double xval = *(lds_cdouble*)(size_t)ad;                                           \
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #T " row_mask:0xf bank_mask:0xf"           \
        : "+v"(acc[(T)&3])                                                                   \
        : "v"(vcur), "v"(xval));                                                             \
  }
  for (; gi < gt.ng; gi += ystep) {
    const int grp = gt.gid[gi];
    const bool more = gi + ystep < gt.ng;
    if (more) fetch(gt.gid[gi + ystep]);       // next group's x rows in flight during the accumulation
    const double ag = cf.alpha[grp], bg = cf.beta[grp];
    const unsigned lane_base = xs_lds + (unsigned)(buf * max_cols * 128) + (unsigned)gc * 8u;
    double* __restrict__ yg = y + (size_t)grp * gsy;
    const double* __restrict__ rg = HAS_R ? r + (size_t)grp * gsr : nullptr;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        if (2 * ch < nhr[rr]) {
          const int lraw = myl[rr][ch];
          const int lcur = lraw & 0x3fffffff;
          const double vcur = fma(ag, mE[rr][ch], ((lraw >> 30) ? bg : 1.0) * mAJ[rr][ch]);
          // the DPP operands are read by hand-written DPP instructions: keep the VALU
          // writes above two wait states away from them (hipcc pads nothing for asm)
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_nop 2");
          __builtin_amdgcn_sched_barrier(0);
          RICADI_FOR8A(RICADI_MS_STEP)
          if (2 * ch + 1 < nhr[rr]) { RICADI_FOR8B(RICADI_MS_STEP) }
        }
      }
      // rows longer than NCH*16 entries: stream the rest
      for (int ch = NCH; 2 * ch < nhr[rr]; ++ch) {
        const int k = ka[rr] + ch * 16 + g;
        int lcur = 0;
        double vcur = 0.0;
        if (k < kb[rr]) {
          const int v = (int)lidx[k];
          lcur = (v & 0x7fff) << 7;
          vcur = fma(ag, vE[k], ((v & 0x8000) ? bg : 1.0) * vAJ[k]);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 2");
        __builtin_amdgcn_sched_barrier(0);
        RICADI_FOR8A(RICADI_MS_STEP)
        if (2 * ch + 1 < nhr[rr]) { RICADI_FOR8B(RICADI_MS_STEP) }
      }
      if (grow[rr] >= 0 && g < m) {
        const int row = grow[rr];
        double out = alpha * ((acc[0] + acc[1]) + (acc[2] + acc[3]));
        if (HAS_R) out += beta_r * rg[(size_t)row * ldr + g];
        yg[(size_t)row * ldy + g] = out;
      }
    }
    if (more) {
      stash(buf ^ 1);
      __syncthreads();                         // tile (buf^1) complete; everybody is done with tile (buf)
      buf ^= 1;
    }
  }
#undef RICADI_MS_STEP
}
size_t spmm_blocked_ms_lds_bytes(int max_cols) { return (size_t)2 * max_cols * 16 * sizeof(double) + 16; }
void launch_spmm_blocked_ms_x32(hipStream_t st, const GroupTab& gt, const double* alphas, const double* betas,
                                int nblk, const int* rows2, const int* rp2, const int* cols2, const uint16_t* lidx,
                                const double* vAJ, const double* vE, const float* x, int ldx, size_t gsx, double* y,
                                int ldy, size_t gsy, double alpha, int m, int max_cols) {
  if (nblk <= 0 || gt.ng <= 0) return;
  GroupCoefs cf;
  for (int i = 0; i < RICADI_MAX_GROUPS; ++i) {
    cf.alpha[i] = alphas[i];
    cf.beta[i] = betas[i];
  }
  int ysplit = 1;
  while (ysplit < gt.ng && (long)nblk * ysplit < 900 && ysplit < 8) ysplit *= 2;
  ysplit = std::min(ysplit, gt.ng);
  const dim3 grid(nblk, ysplit, 1), block(256);
  hipLaunchKernelGGL((spmm_blocked_ms_kernel<false, float>), grid, block, spmm_blocked_ms_lds_bytes(max_cols), st,
                     rows2, rp2, cols2, lidx, gt, cf, vAJ, vE, x, ldx, gsx, y, ldy, gsy, (const double*)nullptr, 0,
                     (size_t)0, alpha, 0.0, m, max_cols);
}
// max_cols <= 160 (tile slots per thread: 16 x XP x XJ), m <= 16, panel offsets in 31 bits
bool spmm_blocked_ms_ok(int m, int max_cols, size_t panel_rows) {
  return m <= 16 && max_cols <= 160 && panel_rows * (size_t)m * 8 < ((size_t)1 << 31);
}
void launch_spmm_blocked_ms(hipStream_t st, const GroupTab& gt, const double* alphas, const double* betas,
                            int nblk, const int* rows2, const int* rp2, const int* cols2,
                            const uint16_t* lidx, const double* vAJ, const double* vE,
                            const double* x, int ldx, size_t gsx, double* y, int ldy, size_t gsy,
                            const double* r, int ldr, size_t gsr, double alpha, double beta_r, int m,
                            int max_cols) {
  if (nblk <= 0 || gt.ng <= 0) return;
  GroupCoefs cf;
  for (int i = 0; i < RICADI_MAX_GROUPS; ++i) {
    cf.alpha[i] = alphas[i];
    cf.beta[i] = betas[i];
  }
  // enough workgroups for ~4 per CU (1024): split the groups over grid.y when the row
  // blocks alone do not fill the chip
  int ysplit = 1;
  while (ysplit < gt.ng && (long)nblk * ysplit < 900 && ysplit < 8) ysplit *= 2;
  static const int ys_env = 0;
  if (ys_env > 0) ysplit = ys_env;
  ysplit = std::min(ysplit, gt.ng);
  const dim3 grid(nblk, ysplit, 1), block(256);
  const size_t lds = spmm_blocked_ms_lds_bytes(max_cols);
  if (r)
    hipLaunchKernelGGL((spmm_blocked_ms_kernel<true>), grid, block, lds, st, rows2, rp2, cols2, lidx, gt,
                       cf, vAJ, vE, x, ldx, gsx, y, ldy, gsy, r, ldr, gsr, alpha, beta_r, m, max_cols);
  else
    hipLaunchKernelGGL((spmm_blocked_ms_kernel<false>), grid, block, lds, st, rows2, rp2, cols2, lidx, gt,
                       cf, vAJ, vE, x, ldx, gsx, y, ldy, gsy, r, ldr, gsr, alpha, beta_r, m, max_cols);
}

// dst[k] = src[perm[k]]  (assembled CSR values -> block order)
__global__ void gather_vals_kernel(int nnz, const int* __restrict__ perm,
                                   const double* __restrict__ src, double* __restrict__ dst) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += gridDim.x * blockDim.x)
    dst[k] = src[perm[k]];
}
void launch_gather_vals(hipStream_t st, int nnz, const int* perm, const double* src, double* dst) {
  int grid = std::min((nnz + 255) / 256, 2048);
  hipLaunchKernelGGL(gather_vals_kernel, dim3(grid), dim3(256), 0, st, nnz, perm, src, dst);
}

// S_val = alpha * srcE + beta * srcA + srcJ on the unified saddle pattern.
__global__ void assemble_shift_kernel(int nnz, const double* __restrict__ srcA,
                                      const double* __restrict__ srcE,
                                      const double* __restrict__ srcJ, double alpha,
                                      double beta, double* __restrict__ out) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += gridDim.x * blockDim.x)
    out[k] = alpha * srcE[k] + beta * srcA[k] + srcJ[k];
}
void launch_assemble_shift(hipStream_t st, int nnz, const double* srcA, const double* srcE,
                           const double* srcJ, double alpha, double beta, double* out) {
  int grid = std::min((nnz + 255) / 256, 2048);
  hipLaunchKernelGGL(assemble_shift_kernel, dim3(grid), dim3(256), 0, st, nnz, srcA, srcE, srcJ,
                     alpha, beta, out);
}


// ---------------------------------------------------------------------------
// elementwise panel helpers (K4)
// ---------------------------------------------------------------------------
__global__ void axpby_kernel(GroupTab gt, size_t n, double a, const double* __restrict__ x,
                             size_t gsx, double b, double* __restrict__ y, size_t gsy) {
  const int grp = gt.gid[blockIdx.z];
  x += (size_t)grp * gsx;
  y += (size_t)grp * gsy;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    y[i] = a * x[i] + (b == 0.0 ? 0.0 : b * y[i]);
}
void launch_axpby_b(hipStream_t st, const GroupTab& gt, size_t n, double a, const double* x,
                    size_t gsx, double b, double* y, size_t gsy) {
  if (!n || gt.ng <= 0) return;
  int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(axpby_kernel, dim3(grid, 1, gt.ng), dim3(256), 0, st, gt, n, a, x, gsx, b, y,
                     gsy);
}
void launch_axpby(hipStream_t st, size_t n, double a, const double* x, double b, double* y) {
  launch_axpby_b(st, single_group(), n, a, x, 0, b, y, 0);
}

// y[r, c] = a[c] * x[r, c] + b * y[r, c]   (per-column scale, contiguous panel)
// yf (optional): FP32 copy of the result; y then holds the SAME rounded values.
template <class LP>
__global__ void colscale_kernel(GroupTab gt, size_t n, int m, const double* __restrict__ a,
                                const double* __restrict__ x, size_t gsx, double b,
                                double* __restrict__ y, size_t gsy, LP* __restrict__ yf,
                                size_t gsf) {
  const int grp = gt.gid[blockIdx.z];
  a += (size_t)grp * m;
  x += (size_t)grp * gsx;
  y += (size_t)grp * gsy;
  if (yf) yf += (size_t)grp * gsf;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    double v = a[i % m] * x[i] + (b == 0.0 ? 0.0 : b * y[i]);
    if (yf) {
      const LP f = (LP)v;
      yf[i] = f;
      v = (double)f;
    }
    y[i] = v;
  }
}
template <class LP>
static void colscale_impl(hipStream_t st, const GroupTab& gt, size_t nrows, int m, const double* a,
                          const double* x, size_t gsx, double b, double* y, size_t gsy, LP* yf,
                          size_t gsf) {
  size_t n = nrows * m;
  if (!n || gt.ng <= 0) return;
  int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(colscale_kernel<LP>, dim3(grid, 1, gt.ng), dim3(256), 0, st, gt, n, m, a, x,
                     gsx, b, y, gsy, yf, gsf);
}
void launch_colscale_b(hipStream_t st, const GroupTab& gt, size_t nrows, int m, const double* a,
                       const double* x, size_t gsx, double b, double* y, size_t gsy, float* yf,
                       size_t gsf) {
  colscale_impl(st, gt, nrows, m, a, x, gsx, b, y, gsy, yf, gsf);
}
void launch_colscale_b(hipStream_t st, const GroupTab& gt, size_t nrows, int m, const double* a,
                       const double* x, size_t gsx, double b, double* y, size_t gsy, _Float16* yf,
                       size_t gsf) {
  colscale_impl(st, gt, nrows, m, a, x, gsx, b, y, gsy, yf, gsf);
}

// copy a strided block of columns: dst[r, dc0 + c] = scale * src[r, sc0 + c], c < w
__global__ void copy_cols_kernel(int nrows, int w, const double* __restrict__ src, int lds_,
                                 int sc0, double* __restrict__ dst, int ldd, int dc0,
                                 double scale) {
  size_t n = (size_t)nrows * w;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i / w;
    int c = (int)(i % w);
    dst[r * ldd + dc0 + c] = scale * src[r * lds_ + sc0 + c];
  }
}
void launch_copy_cols(hipStream_t st, int nrows, int w, const double* src, int lds_, int sc0,
                      double* dst, int ldd, int dc0, double scale) {
  size_t n = (size_t)nrows * w;
  if (!n) return;
  int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(copy_cols_kernel, dim3(grid), dim3(256), 0, st, nrows, w, src, lds_, sc0, dst,
                     ldd, dc0, scale);
}

// ---------------------------------------------------------------------------
// K3: per-column Krylov orthogonalisation.
//
// cols_dots: partial[blk][i][c] = sum_{r in chunk} V_i[r,c] * w[r,c]  for
// i < nvec (vector nvec, if want_self, is w itself -> ||w||^2 per column).
// The w chunk is staged in LDS once and re-used against every basis panel
// (the "LDS-staged Krylov panel"); a thread owns one (i, c) output, the 16
// lanes of a group read one contiguous row of V_i, so no cross-lane reduction
// is needed at all.  A second kernel sums the partials over workgroups.
// ---------------------------------------------------------------------------
constexpr int DOT_ROWS = 64;

template <class BT>
__global__ __launch_bounds__(256) void cols_dots_kernel(
    GroupTab gt, int nrows, int m, int nvec, const BT* __restrict__ basis, size_t vstride,
    size_t gsb, const double* __restrict__ w, size_t gsw, int want_self,
    double* __restrict__ partial, size_t gsp) {
  extern __shared__ double wl[];  // DOT_ROWS x m
  const int grp = gt.gid[blockIdx.z];
  basis += (size_t)grp * gsb;
  w += (size_t)grp * gsw;
  partial += (size_t)grp * gsp;
  const int r0 = blockIdx.x * DOT_ROWS;
  const int nr = min(DOT_ROWS, nrows - r0);
  for (int e = threadIdx.x; e < nr * m; e += blockDim.x) wl[e] = w[(size_t)r0 * m + e];
  __syncthreads();
  const int ntot = nvec + (want_self ? 1 : 0);
  const int nout = ntot * m;
  for (int o = threadIdx.x; o < nout; o += blockDim.x) {
    const int i = o / m, c = o - i * m;
    double s0 = 0.0, s1 = 0.0;
    int r = 0;
    if (i < nvec) {
      const BT* v = basis + (size_t)i * vstride + (size_t)r0 * m + c;
      double s2 = 0.0, s3 = 0.0;
      for (; r + 3 < nr; r += 4) {       // four independent row loads in flight
        const double v0 = (double)v[(size_t)r * m], v1 = (double)v[(size_t)(r + 1) * m];
        const double v2 = (double)v[(size_t)(r + 2) * m], v3 = (double)v[(size_t)(r + 3) * m];
        s0 = fma(v0, wl[r * m + c], s0);
        s1 = fma(v1, wl[(r + 1) * m + c], s1);
        s2 = fma(v2, wl[(r + 2) * m + c], s2);
        s3 = fma(v3, wl[(r + 3) * m + c], s3);
      }
      for (; r < nr; ++r) s0 = fma((double)v[(size_t)r * m], wl[r * m + c], s0);
      s0 += s2;
      s1 += s3;
    } else {
      for (; r < nr; ++r) s0 = fma(wl[r * m + c], wl[r * m + c], s0);
    }
    partial[(size_t)blockIdx.x * nout + o] = s0 + s1;
  }
}

// out[o] (+)= sum_b partial[b][o].  256 threads = 16 outputs x 16 block-slices:
// the 16 lanes of a group read 16 consecutive outputs of one partial row (one
// 128-B line), the 16 groups stride over the workgroups; LDS tree at the end.
__global__ __launch_bounds__(256) void reduce_partials_kernel(GroupTab gt, int nblk, int nout,
                                                              const double* __restrict__ partial,
                                                              size_t gsp, double* __restrict__ out,
                                                              size_t gso, int accumulate) {
  __shared__ double red[16][17];
  const int grp = gt.gid[blockIdx.z];
  partial += (size_t)grp * gsp;
  out += (size_t)grp * gso;
  const int oo = threadIdx.x & 15, bsl = threadIdx.x >> 4;
  const int o = blockIdx.x * 16 + oo;
  double s0 = 0.0, s1 = 0.0;
  if (o < nout) {
    // eight loads in flight per thread (two left the kernel waiting on ~15 dependent round trips: 6.6 us)
    int b = bsl;
    double t[8];
    for (; b + 112 < nblk; b += 128) {
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = partial[(size_t)(b + 16 * u) * nout + o];
      s0 += (t[0] + t[2]) + (t[4] + t[6]);
      s1 += (t[1] + t[3]) + (t[5] + t[7]);
    }
    for (; b + 16 < nblk; b += 32) {
      s0 += partial[(size_t)b * nout + o];
      s1 += partial[(size_t)(b + 16) * nout + o];
    }
    if (b < nblk) s0 += partial[(size_t)b * nout + o];
  }
  red[bsl][oo] = s0 + s1;
  __syncthreads();
  if (bsl == 0 && o < nout) {
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += red[t][oo];
    out[o] = accumulate ? out[o] + s : s;
  }
}

int dots_num_blocks(int nrows) { return (nrows + DOT_ROWS - 1) / DOT_ROWS; }

// ---------------------------------------------------------------------------
// K3, FP16-stored basis with 16-column panels (the hot case): the same three Arnoldi passes
// with 16-byte (8 x FP16) basis loads.  The generic kernels above read 2 bytes per lane and
// load, which is fine while the launches are latency bound (n ~ 3e4) and leaves them at
// 0.34-0.46 of the HBM roofline at n = 5e5.
//   dots: lane = (row slice s = lane & 15, column half, vector) -- the 16 lanes of a DPP row
//   hold the 16 row slices of ONE (vector, half), each lane runs over rows s, s+16, s+32, s+48
//   of the 64-row chunk with its 4 loads in flight, and the row sum is 4 DPP exchanges.
// ---------------------------------------------------------------------------
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ double dpp_xchg(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a DPP row, result in every lane
__device__ __forceinline__ double dpp_row_sum(double v) {
  v += dpp_xchg<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_xchg<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_xchg<0x141>(v);   // row_half_mirror
  v += dpp_xchg<0x140>(v);   // row_mirror
  return v;
}

// LDS row stride of the staged chunk: 18 doubles (144 B) -- with 16 the lanes of a DPP row (rows s, s+1, ...
// 128 B apart) fall on two banks sets and every read is an 8-way conflict
constexpr int WLS = 18;
// dot products of the chunk held in wl (DOT_ROWS rows of 16 doubles, stride WLS; rows >= nr zeroed) against the
// basis vectors [0, nvec) and, if want_self, against itself (output row nvec)
template <bool ATOMIC = false>
__device__ __forceinline__ void chunk_dots16(const _Float16* __restrict__ basis, size_t vstride, int r0, int nr,
                                             int nvec, int want_self, const double* wl,
                                             double* __restrict__ pout) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = lane & 15, half = (lane >> 4) & 1, vsub = lane >> 5;
  const int ntot = nvec + (want_self ? 1 : 0);
  for (int i0 = 0; i0 < ntot; i0 += 8) {
    const int i = i0 + 2 * wave + vsub;
    double acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = 0.0;
    if (i < nvec) {
      const _Float16* v = basis + (size_t)i * vstride + (size_t)r0 * 16 + half * 8;
      half8_t x[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = s + 16 * k;
        if (row < nr) x[k] = *reinterpret_cast<const half8_t*>(v + (size_t)row * 16);
        else x[k] = (half8_t)(_Float16)0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double* wr = wl + (s + 16 * k) * WLS + half * 8;
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = fma((double)x[k][t], wr[t], acc[t]);
      }
    } else if (i == nvec && want_self) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double* wr = wl + (s + 16 * k) * WLS + half * 8;
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = fma(wr[t], wr[t], acc[t]);
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = dpp_row_sum(acc[t]);
    if (s == 0 && i < ntot) {
      if (ATOMIC) {
        // straight into the (pre-zeroed) result: no partial rows, no reduce launch
#pragma unroll
        for (int t = 0; t < 8; ++t) atomicAdd(pout + (size_t)i * 16 + half * 8 + t, acc[t]);
      } else {
        double2* o = reinterpret_cast<double2*>(pout + (size_t)i * 16 + half * 8);
        o[0] = make_double2(acc[0], acc[1]);
        o[1] = make_double2(acc[2], acc[3]);
        o[2] = make_double2(acc[4], acc[5]);
        o[3] = make_double2(acc[6], acc[7]);
      }
    }
  }
}

// ATOMIC: `partial` is the result array itself (group stride gsp), zeroed beforehand
template <bool ATOMIC = false>
__global__ __launch_bounds__(256) void cols_dots16_kernel(
    GroupTab gt, int nrows, int nvec, const _Float16* __restrict__ basis, size_t vstride, size_t gsb,
    const double* __restrict__ w, size_t gsw, int want_self, double* __restrict__ partial, size_t gsp) {
  __shared__ __attribute__((aligned(16))) double wl[DOT_ROWS * WLS];
  const int grp = gt.gid[blockIdx.z];
  basis += (size_t)grp * gsb;
  w += (size_t)grp * gsw;
  partial += (size_t)grp * gsp;
  const int r0 = blockIdx.x * DOT_ROWS;
  const int nr = min(DOT_ROWS, nrows - r0);
  {
    const double2* src = reinterpret_cast<const double2*>(w + (size_t)r0 * 16);
    for (int e = threadIdx.x; e < DOT_ROWS * 8; e += 256)
      *reinterpret_cast<double2*>(wl + (e >> 3) * WLS + (e & 7) * 2) = e < nr * 8 ? src[e] : make_double2(0.0, 0.0);
  }
  __syncthreads();
  const int nout = (nvec + (want_self ? 1 : 0)) * 16;
  chunk_dots16<ATOMIC>(basis, vstride, r0, nr, nvec, want_self, wl, ATOMIC ? partial : partial + (size_t)blockIdx.x * nout);
}

// w' = w - V h (written back), then the dots of w' against V and itself (chunk_dots16; the basis chunk
// is cache resident by then, so the LDS side decides: with unpadded rows this phase was 2x slower)
// STORE = false: w' is only staged in LDS for the dots, the panel w keeps the vector BEFORE the first projection (the
// final update then subtracts the basis with the SUM of both passes' coefficients: one 8-byte store per element less)
template <bool ATOMIC = false, bool STORE = true>
__global__ __launch_bounds__(256) void cols_update_dots16_kernel(
    GroupTab gt, int nrows, int nvec, const _Float16* __restrict__ basis, size_t vstride, size_t gsb,
    const double* __restrict__ h, size_t gsh, double* __restrict__ w, size_t gsw,
    double* __restrict__ partial, size_t gsp) {
  extern __shared__ __attribute__((aligned(16))) double sm16[];
  double* wl = sm16;                       // DOT_ROWS rows, stride WLS
  double* hl = sm16 + DOT_ROWS * WLS;      // nvec x 16
  const int grp = gt.gid[blockIdx.z];
  basis += (size_t)grp * gsb;
  h += (size_t)grp * gsh;
  w += (size_t)grp * gsw;
  partial += (size_t)grp * gsp;
  const int r0 = blockIdx.x * DOT_ROWS;
  const int nr = min(DOT_ROWS, nrows - r0);
  for (int e = threadIdx.x; e < nvec * 16; e += 256) hl[e] = h[e];
  __syncthreads();
  {
    // update: a thread owns the elements tid, tid + 256, ... of the chunk (one column c = tid & 15, four rows);
    // per pair of basis vectors its 8 two-byte loads are issued together and the two coefficients come from
    // LDS once.  (The 16-byte form with the vectors split over lane pairs was slower at every basis size:
    // 1.69 vs 1.02 ms at n = 5e5, 7 vectors.)
    const size_t base = (size_t)r0 * 16;
    const int c = threadIdx.x & 15;
    constexpr int NE = DOT_ROWS * 16 / 256;          // 4
    int e[NE];
    bool ok[NE];
    double sacc[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      e[k] = threadIdx.x + 256 * k;
      ok[k] = e[k] < nr * 16;
      sacc[k] = 0.0;
    }
    int i = 0;
    for (; i + 1 < nvec; i += 2) {
      _Float16 b0[NE], b1[NE];
      const _Float16* v0 = basis + (size_t)i * vstride + base;
      const _Float16* v1 = v0 + vstride;
#pragma unroll
      for (int k = 0; k < NE; ++k) {
        b0[k] = v0[ok[k] ? e[k] : 0];               // unconditional loads (row 0 of the chunk is always valid)
        b1[k] = v1[ok[k] ? e[k] : 0];
      }
      const double h0 = hl[i * 16 + c], h1 = hl[(i + 1) * 16 + c];
#pragma unroll
      for (int k = 0; k < NE; ++k) sacc[k] = fma(h1, (double)b1[k], fma(h0, (double)b0[k], sacc[k]));
    }
    if (i < nvec) {
      _Float16 b0[NE];
      const _Float16* v0 = basis + (size_t)i * vstride + base;
#pragma unroll
      for (int k = 0; k < NE; ++k) b0[k] = v0[ok[k] ? e[k] : 0];
      const double h0 = hl[i * 16 + c];
#pragma unroll
      for (int k = 0; k < NE; ++k) sacc[k] = fma(h0, (double)b0[k], sacc[k]);
    }
    double wv[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) wv[k] = w[base + (ok[k] ? e[k] : 0)];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const double v = ok[k] ? wv[k] - sacc[k] : 0.0;
      if (STORE && ok[k]) w[base + e[k]] = v;
      wl[(e[k] >> 4) * WLS + c] = v;
    }
  }
  __syncthreads();
  chunk_dots16<ATOMIC>(basis, vstride, r0, nr, nvec, 1, wl,
                       ATOMIC ? partial : partial + (size_t)blockIdx.x * (nvec + 1) * 16);
}

// ---- the same two dot kernels for panels of 8 * NOCT columns (NOCT = 1, 3, 4: the projection solve, the
// augmented Sherman-Morrison-Woodbury sweep [b, U] of the Newton step and wider panels): a DPP row holds the 16 row
// slices of one (vector, column octet) pair.
template <int NOCT>
__device__ __forceinline__ void chunk_dots8x(const _Float16* __restrict__ basis, size_t vstride, int r0, int nr,
                                             int nvec, int want_self, const double* wl, double* __restrict__ pout) {
  constexpr int M = 8 * NOCT, WS = M + 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = lane & 15, pr = lane >> 4;
  const int ntot = nvec + (want_self ? 1 : 0);
  const int npairs = ntot * NOCT;
  for (int pq0 = 0; pq0 < npairs; pq0 += 16) {
    const int pq = pq0 + 4 * wave + pr;
    const int i = pq / NOCT, o = pq - i * NOCT;
    double acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = 0.0;
    if (i < nvec) {
      const _Float16* v = basis + (size_t)i * vstride + (size_t)r0 * M + o * 8;
      half8_t x[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = s + 16 * k;
        if (row < nr) x[k] = *reinterpret_cast<const half8_t*>(v + (size_t)row * M);
        else x[k] = (half8_t)(_Float16)0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double* wr = wl + (s + 16 * k) * WS + o * 8;
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = fma((double)x[k][t], wr[t], acc[t]);
      }
    } else if (i == nvec && want_self) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double* wr = wl + (s + 16 * k) * WS + o * 8;
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = fma(wr[t], wr[t], acc[t]);
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = dpp_row_sum(acc[t]);
    if (s == 0 && i < ntot) {
      double2* op = reinterpret_cast<double2*>(pout + (size_t)i * M + o * 8);
      op[0] = make_double2(acc[0], acc[1]);
      op[1] = make_double2(acc[2], acc[3]);
      op[2] = make_double2(acc[4], acc[5]);
      op[3] = make_double2(acc[6], acc[7]);
    }
  }
}
template <int NOCT>
__global__ __launch_bounds__(256) void cols_dots8x_kernel(
    GroupTab gt, int nrows, int nvec, const _Float16* __restrict__ basis, size_t vstride, size_t gsb,
    const double* __restrict__ w, size_t gsw, int want_self, double* __restrict__ partial, size_t gsp) {
  constexpr int M = 8 * NOCT, WS = M + 2;
  __shared__ __attribute__((aligned(16))) double wl[DOT_ROWS * WS];
  const int grp = gt.gid[blockIdx.z];
  basis += (size_t)grp * gsb;
  w += (size_t)grp * gsw;
  partial += (size_t)grp * gsp;
  const int r0 = blockIdx.x * DOT_ROWS;
  const int nr = min(DOT_ROWS, nrows - r0);
  {
    const double2* src = reinterpret_cast<const double2*>(w + (size_t)r0 * M);
    for (int e = threadIdx.x; e < DOT_ROWS * (M / 2); e += 256) {
      const int row = e / (M / 2), c2 = e - row * (M / 2);
      *reinterpret_cast<double2*>(wl + row * WS + 2 * c2) = row < nr ? src[e] : make_double2(0.0, 0.0);
    }
  }
  __syncthreads();
  const int nout = (nvec + (want_self ? 1 : 0)) * M;
  chunk_dots8x<NOCT>(basis, vstride, r0, nr, nvec, want_self, wl, partial + (size_t)blockIdx.x * nout);
}
template <int NOCT>
__global__ __launch_bounds__(256) void cols_update_dots8x_kernel(
    GroupTab gt, int nrows, int nvec, const _Float16* __restrict__ basis, size_t vstride, size_t gsb,
    const double* __restrict__ h, size_t gsh, double* __restrict__ w, size_t gsw,
    double* __restrict__ partial, size_t gsp) {
  constexpr int M = 8 * NOCT, WS = M + 2;
  extern __shared__ __attribute__((aligned(16))) double sm8x[];
  double* wl = sm8x;                       // DOT_ROWS rows, stride WS
  double* hl = sm8x + DOT_ROWS * WS;       // nvec x M
  const int grp = gt.gid[blockIdx.z];
  basis += (size_t)grp * gsb;
  h += (size_t)grp * gsh;
  w += (size_t)grp * gsw;
  partial += (size_t)grp * gsp;
  const int r0 = blockIdx.x * DOT_ROWS;
  const int nr = min(DOT_ROWS, nrows - r0);
  for (int e = threadIdx.x; e < nvec * M; e += 256) hl[e] = h[e];
  __syncthreads();
  {
    const size_t base = (size_t)r0 * M;
    constexpr int NE = DOT_ROWS * M / 256;           // 2 * NOCT
    int e[NE], c[NE];
    bool ok[NE];
    double sacc[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      e[k] = threadIdx.x + 256 * k;
      c[k] = e[k] % M;
      ok[k] = e[k] < nr * M;
      sacc[k] = 0.0;
    }
    for (int i = 0; i < nvec; ++i) {
      _Float16 b0[NE];
      const _Float16* v0 = basis + (size_t)i * vstride + base;
#pragma unroll
      for (int k = 0; k < NE; ++k) b0[k] = v0[ok[k] ? e[k] : 0];
#pragma unroll
      for (int k = 0; k < NE; ++k) sacc[k] = fma(hl[i * M + c[k]], (double)b0[k], sacc[k]);
    }
    double wv[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) wv[k] = w[base + (ok[k] ? e[k] : 0)];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const double v = ok[k] ? wv[k] - sacc[k] : 0.0;
      if (ok[k]) w[base + e[k]] = v;
      wl[(e[k] / M) * WS + c[k]] = v;
    }
  }
  __syncthreads();
  chunk_dots8x<NOCT>(basis, vstride, r0, nr, nvec, 1, wl, partial + (size_t)blockIdx.x * (nvec + 1) * M);
}

// out = scale * (w + sign * V h), stored in FP16 (outf) and, rounded identically, in FP64 (out):
// thread = (row, column half), 16-byte basis loads, four vectors in flight
__global__ __launch_bounds__(256) void cols_update16_kernel(
    GroupTab gt, size_t nhalf, GroupInts nvecs, const _Float16* __restrict__ basis, size_t vstride, size_t gsb,
    const double* __restrict__ h, size_t gsh, double sign, const double* __restrict__ w, size_t gsw,
    const double* __restrict__ scale, double* __restrict__ out, size_t gso, _Float16* __restrict__ outf,
    size_t gsf, int m) {
  extern __shared__ double hl[];           // nvec x m  (m = 8, 16, 24 or 32 columns)
  const int noct = m >> 3;
  const int grp = gt.gid[blockIdx.z];
  const int nvec = nvecs.v[grp];
  basis += (size_t)grp * gsb;
  h += (size_t)grp * gsh;
  if (w) w += (size_t)grp * gsw;
  if (scale) scale += (size_t)grp * m;
  if (out) out += (size_t)grp * gso;
  if (outf) outf += (size_t)grp * gsf;
  for (int e = threadIdx.x; e < nvec * m; e += 256) hl[e] = h[e];
  __syncthreads();
  for (size_t idx = blockIdx.x * (size_t)256 + threadIdx.x; idx < nhalf; idx += (size_t)gridDim.x * 256) {
    const size_t e = idx * 8;
    const int c0 = (int)(idx % (size_t)noct) * 8;
    double a[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) a[t] = 0.0;
    const _Float16* v = basis + e;
    int i = 0;
    for (; i + 3 < nvec; i += 4) {
      half8_t x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const half8_t*>(v + (size_t)(i + u) * vstride);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 8; ++t) a[t] = fma(hl[(i + u) * m + c0 + t], (double)x[u][t], a[t]);
    }
    for (; i < nvec; ++i) {
      const half8_t x = *reinterpret_cast<const half8_t*>(v + (size_t)i * vstride);
#pragma unroll
      for (int t = 0; t < 8; ++t) a[t] = fma(hl[i * m + c0 + t], (double)x[t], a[t]);
    }
    if (w) {
      const double2* wp = reinterpret_cast<const double2*>(w + e);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double2 ww = wp[t];
        a[2 * t] = ww.x + sign * a[2 * t];
        a[2 * t + 1] = ww.y + sign * a[2 * t + 1];
      }
    } else {
#pragma unroll
      for (int t = 0; t < 8; ++t) a[t] *= sign;
    }
    if (scale) {
#pragma unroll
      for (int t = 0; t < 8; ++t) a[t] *= scale[c0 + t];
    }
    if (outf) {
      half8_t f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        f[t] = (_Float16)a[t];
        a[t] = (double)f[t];
      }
      *reinterpret_cast<half8_t*>(outf + e) = f;
    }
    if (out) {
      double2* op = reinterpret_cast<double2*>(out + e);
#pragma unroll
      for (int t = 0; t < 4; ++t) op[t] = make_double2(a[2 * t], a[2 * t + 1]);
    }
  }
}
// Same update for an FP32-stored basis (the Z_j of the flexible GMRES: the correction x += Z y at the end of a
// restart cycle): thread = 4 consecutive elements, 16-byte loads, four vectors in flight.  No stored copy.
__global__ __launch_bounds__(256) void cols_update_f4_kernel(
    GroupTab gt, size_t nquad, GroupInts nvecs, const float* __restrict__ basis, size_t vstride, size_t gsb,
    const double* __restrict__ h, size_t gsh, double sign, const double* __restrict__ w, size_t gsw,
    const double* __restrict__ scale, double* __restrict__ out, size_t gso, int m) {
  extern __shared__ double hl[];           // nvec x m
  const int grp = gt.gid[blockIdx.z];
  const int nvec = nvecs.v[grp];
  basis += (size_t)grp * gsb;
  h += (size_t)grp * gsh;
  if (w) w += (size_t)grp * gsw;
  if (scale) scale += (size_t)grp * m;
  out += (size_t)grp * gso;
  for (int e = threadIdx.x; e < nvec * m; e += 256) hl[e] = h[e];
  __syncthreads();
  for (size_t idx = blockIdx.x * (size_t)256 + threadIdx.x; idx < nquad; idx += (size_t)gridDim.x * 256) {
    const size_t e = idx * 4;
    const int c0 = (int)(e % (size_t)m);
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    const float* v = basis + e;
    int i = 0;
    for (; i + 3 < nvec; i += 4) {
      float4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const float4*>(v + (size_t)(i + u) * vstride);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double* hh = hl + (i + u) * m + c0;
        a[0] = fma(hh[0], (double)x[u].x, a[0]);
        a[1] = fma(hh[1], (double)x[u].y, a[1]);
        a[2] = fma(hh[2], (double)x[u].z, a[2]);
        a[3] = fma(hh[3], (double)x[u].w, a[3]);
      }
    }
    for (; i < nvec; ++i) {
      const float4 x = *reinterpret_cast<const float4*>(v + (size_t)i * vstride);
      const double* hh = hl + i * m + c0;
      a[0] = fma(hh[0], (double)x.x, a[0]);
      a[1] = fma(hh[1], (double)x.y, a[1]);
      a[2] = fma(hh[2], (double)x.z, a[2]);
      a[3] = fma(hh[3], (double)x.w, a[3]);
    }
    if (w) {
      const double2* wp = reinterpret_cast<const double2*>(w + e);
      const double2 w0 = wp[0], w1 = wp[1];
      a[0] = w0.x + sign * a[0];
      a[1] = w0.y + sign * a[1];
      a[2] = w1.x + sign * a[2];
      a[3] = w1.y + sign * a[3];
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] *= sign;
    }
    if (scale) {
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] *= scale[c0 + t];
    }
    double2* op = reinterpret_cast<double2*>(out + e);
    op[0] = make_double2(a[0], a[1]);
    op[1] = make_double2(a[2], a[3]);
  }
}
// the launch classes that use these kernels (1 dots, 2 update+dots, 4 update; 8: also for panels of 8, 24 and 32 columns)
static bool arnoldi16(int) { return true; }

template <class BT>
static void cols_dots_impl(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                           const BT* basis, size_t vstride, size_t gsb, const double* w, size_t gsw,
                           int want_self, double* partial, size_t gsp, double* out, size_t gso) {
  const int nblk = dots_num_blocks(nrows);
  const int nout = (nvec + (want_self ? 1 : 0)) * m;
  if (nout == 0 || gt.ng <= 0) return;
  if constexpr (std::is_same<BT, _Float16>::value) {
    if ((m == 8 || m == 24 || m == 32) && arnoldi16(1) && arnoldi16(8)) {
      const dim3 grid(nblk, 1, gt.ng);
      if (m == 8)
        hipLaunchKernelGGL((cols_dots8x_kernel<1>), grid, dim3(256), 0, st, gt, nrows, nvec, basis, vstride, gsb, w, gsw,
                           want_self, partial, gsp);
      else if (m == 24)
        hipLaunchKernelGGL((cols_dots8x_kernel<3>), grid, dim3(256), 0, st, gt, nrows, nvec, basis, vstride, gsb, w, gsw,
                           want_self, partial, gsp);
      else
        hipLaunchKernelGGL((cols_dots8x_kernel<4>), grid, dim3(256), 0, st, gt, nrows, nvec, basis, vstride, gsb, w, gsw,
                           want_self, partial, gsp);
      hipLaunchKernelGGL(reduce_partials_kernel, dim3((nout + 15) / 16, 1, gt.ng), dim3(256), 0, st, gt,
                         nblk, nout, partial, gsp, out, gso, 0);
      return;
    }
    if (m == 16 && arnoldi16(1)) {
      hipLaunchKernelGGL(cols_dots16_kernel<false>, dim3(nblk, 1, gt.ng), dim3(256), 0, st, gt, nrows, nvec, basis,
                         vstride, gsb, w, gsw, want_self, partial, gsp);
      hipLaunchKernelGGL(reduce_partials_kernel, dim3((nout + 15) / 16, 1, gt.ng), dim3(256), 0, st, gt,
                         nblk, nout, partial, gsp, out, gso, 0);
      return;
    }
  }
  hipLaunchKernelGGL(cols_dots_kernel<BT>, dim3(nblk, 1, gt.ng), dim3(256),
                     DOT_ROWS * m * sizeof(double), st, gt, nrows, m, nvec, basis, vstride, gsb, w,
                     gsw, want_self, partial, gsp);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((nout + 15) / 16, 1, gt.ng), dim3(256), 0, st, gt,
                     nblk, nout, partial, gsp, out, gso, 0);
}
void launch_cols_dots_b(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                        const double* basis, size_t vstride, size_t gsb, const double* w, size_t gsw,
                        int want_self, double* partial, size_t gsp, double* out, size_t gso) {
  cols_dots_impl(st, gt, nrows, m, nvec, basis, vstride, gsb, w, gsw, want_self, partial, gsp, out,
                 gso);
}
void launch_cols_dots_b(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                        const float* basis, size_t vstride, size_t gsb, const double* w, size_t gsw,
                        int want_self, double* partial, size_t gsp, double* out, size_t gso) {
  cols_dots_impl(st, gt, nrows, m, nvec, basis, vstride, gsb, w, gsw, want_self, partial, gsp, out,
                 gso);
}
void launch_cols_dots_b(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                        const _Float16* basis, size_t vstride, size_t gsb, const double* w,
                        size_t gsw, int want_self, double* partial, size_t gsp, double* out,
                        size_t gso) {
  cols_dots_impl(st, gt, nrows, m, nvec, basis, vstride, gsb, w, gsw, want_self, partial, gsp, out,
                 gso);
}
// (The two dot passes with FP64 atomic accumulation instead of partial rows + reduce launches -- two launches fewer per
// iteration -- were measured in round 3: 468 workgroups per group add to the same 112-192 addresses, contended FP64 atomics
// serialise at the memory side, cfg2 step 393 -> 741 ms.  The launchers are gone; the kernels keep their ATOMIC template
// parameter at false.)
void launch_cols_dots(hipStream_t st, int nrows, int m, int nvec, const double* basis,
                      size_t vstride, const double* w, int want_self, double* partial,
                      double* out) {
  launch_cols_dots_b(st, single_group(), nrows, m, nvec, basis, vstride, 0, w, 0, want_self, partial,
                     0, out, 0);
}

// Fused CGS2 middle step: for a chunk of DOT_ROWS rows
//   w'[r,c] = w[r,c] - sum_i h[i,c] V_i[r,c]          (first Gram-Schmidt update, written back)
//   partial[blk][i,c] = sum_r V_i[r,c] w'[r,c],  i <= nvec  (row nvec = ||w'||^2)
// The dot products of the second pass are row-local, so the block computes them
// right after its slice of w' -- the basis slice it has just read is still in
// L1/L2 -- which saves one launch and one pass over the Krylov basis per
// iteration compared with separate update and dots kernels.
template <class BT>
__global__ __launch_bounds__(256) void cols_update_dots_kernel(
    GroupTab gt, int nrows, int m, int nvec, const BT* __restrict__ basis, size_t vstride,
    size_t gsb, const double* __restrict__ h, size_t gsh, double* __restrict__ w, size_t gsw,
    double* __restrict__ partial, size_t gsp) {
  extern __shared__ double wl[];  // DOT_ROWS x m
  const int grp = gt.gid[blockIdx.z];
  basis += (size_t)grp * gsb;
  h += (size_t)grp * gsh;
  w += (size_t)grp * gsw;
  partial += (size_t)grp * gsp;
  const int r0 = blockIdx.x * DOT_ROWS;
  const int nr = min(DOT_ROWS, nrows - r0);
  const size_t base = (size_t)r0 * m;
  for (int e = threadIdx.x; e < nr * m; e += blockDim.x) {
    const int c = e % m;
    double s0 = 0.0, s1 = 0.0;
    int i = 0;
    for (; i + 1 < nvec; i += 2) {
      s0 = fma(h[i * m + c], (double)basis[(size_t)i * vstride + base + e], s0);
      s1 = fma(h[(i + 1) * m + c], (double)basis[(size_t)(i + 1) * vstride + base + e], s1);
    }
    if (i < nvec) s0 = fma(h[i * m + c], (double)basis[(size_t)i * vstride + base + e], s0);
    const double v = w[base + e] - (s0 + s1);
    wl[e] = v;
    w[base + e] = v;
  }
  __syncthreads();
  const int nout = (nvec + 1) * m;
  for (int o = threadIdx.x; o < nout; o += blockDim.x) {
    const int i = o / m, c = o - i * m;
    double s0 = 0.0, s1 = 0.0;
    if (i < nvec) {
      const BT* v = basis + (size_t)i * vstride + base + c;
      int r = 0;
      double s2 = 0.0, s3 = 0.0;
      for (; r + 3 < nr; r += 4) {
        const double v0 = (double)v[(size_t)r * m], v1 = (double)v[(size_t)(r + 1) * m];
        const double v2 = (double)v[(size_t)(r + 2) * m], v3 = (double)v[(size_t)(r + 3) * m];
        s0 = fma(v0, wl[r * m + c], s0);
        s1 = fma(v1, wl[(r + 1) * m + c], s1);
        s2 = fma(v2, wl[(r + 2) * m + c], s2);
        s3 = fma(v3, wl[(r + 3) * m + c], s3);
      }
      for (; r < nr; ++r) s0 = fma((double)v[(size_t)r * m], wl[r * m + c], s0);
      s0 += s2;
      s1 += s3;
    } else {
      for (int r = 0; r < nr; ++r) s0 = fma(wl[r * m + c], wl[r * m + c], s0);
    }
    partial[(size_t)blockIdx.x * nout + o] = s0 + s1;
  }
}
// set by update_dots_keeps_w(): the 16-column FP16 launch leaves w untouched (see cols_update_dots16_kernel)
static thread_local bool g_update_dots_nostore = false;
bool update_dots_keeps_w(int m, bool fp16_basis, int nvec_max) {
  return fp16_basis && m == 16 && arnoldi16(2) && arnoldi16(4) &&
         (size_t)(DOT_ROWS * 18 + nvec_max * 16) * sizeof(double) <= 48 * 1024;
}
void set_update_dots_nostore(bool v) { g_update_dots_nostore = v; }
template <class BT>
static void cols_update_dots_impl(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                                  const BT* basis, size_t vstride, size_t gsb, const double* h,
                                  size_t gsh, double* w, size_t gsw, double* partial, size_t gsp,
                                  double* out, size_t gso) {
  if (gt.ng <= 0) return;
  const int nblk = dots_num_blocks(nrows);
  const int nout = (nvec + 1) * m;
  if constexpr (std::is_same<BT, _Float16>::value) {
    if ((m == 8 || m == 24 || m == 32) && arnoldi16(2) && arnoldi16(8) &&
        (size_t)(DOT_ROWS * (m + 2) + nvec * m) * sizeof(double) <= 48 * 1024) {
      const dim3 grid(nblk, 1, gt.ng);
      const size_t lds = (size_t)(DOT_ROWS * (m + 2) + nvec * m) * sizeof(double);
      if (m == 8)
        hipLaunchKernelGGL((cols_update_dots8x_kernel<1>), grid, dim3(256), lds, st, gt, nrows, nvec, basis, vstride,
                           gsb, h, gsh, w, gsw, partial, gsp);
      else if (m == 24)
        hipLaunchKernelGGL((cols_update_dots8x_kernel<3>), grid, dim3(256), lds, st, gt, nrows, nvec, basis, vstride,
                           gsb, h, gsh, w, gsw, partial, gsp);
      else
        hipLaunchKernelGGL((cols_update_dots8x_kernel<4>), grid, dim3(256), lds, st, gt, nrows, nvec, basis, vstride,
                           gsb, h, gsh, w, gsw, partial, gsp);
      hipLaunchKernelGGL(reduce_partials_kernel, dim3((nout + 15) / 16, 1, gt.ng), dim3(256), 0, st, gt,
                         nblk, nout, partial, gsp, out, gso, 0);
      return;
    }
    if (m == 16 && arnoldi16(2) && (size_t)(DOT_ROWS * 18 + nvec * 16) * sizeof(double) <= 48 * 1024) {
      if (g_update_dots_nostore)
        hipLaunchKernelGGL((cols_update_dots16_kernel<false, false>), dim3(nblk, 1, gt.ng), dim3(256),
                           (size_t)(DOT_ROWS * 18 + nvec * 16) * sizeof(double), st, gt, nrows, nvec, basis, vstride,
                           gsb, h, gsh, w, gsw, partial, gsp);
      else
        hipLaunchKernelGGL((cols_update_dots16_kernel<false, true>), dim3(nblk, 1, gt.ng), dim3(256),
                           (size_t)(DOT_ROWS * 18 + nvec * 16) * sizeof(double), st, gt, nrows, nvec, basis, vstride,
                           gsb, h, gsh, w, gsw, partial, gsp);
      hipLaunchKernelGGL(reduce_partials_kernel, dim3((nout + 15) / 16, 1, gt.ng), dim3(256), 0, st, gt,
                         nblk, nout, partial, gsp, out, gso, 0);
      return;
    }
  }
  hipLaunchKernelGGL(cols_update_dots_kernel<BT>, dim3(nblk, 1, gt.ng), dim3(256),
                     DOT_ROWS * m * sizeof(double), st, gt, nrows, m, nvec, basis, vstride, gsb, h,
                     gsh, w, gsw, partial, gsp);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((nout + 15) / 16, 1, gt.ng), dim3(256), 0, st, gt,
                     nblk, nout, partial, gsp, out, gso, 0);
}
void launch_cols_update_dots_b(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                               const double* basis, size_t vstride, size_t gsb, const double* h,
                               size_t gsh, double* w, size_t gsw, double* partial, size_t gsp,
                               double* out, size_t gso) {
  cols_update_dots_impl(st, gt, nrows, m, nvec, basis, vstride, gsb, h, gsh, w, gsw, partial, gsp, out,
                        gso);
}
void launch_cols_update_dots_b(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                               const float* basis, size_t vstride, size_t gsb, const double* h,
                               size_t gsh, double* w, size_t gsw, double* partial, size_t gsp,
                               double* out, size_t gso) {
  cols_update_dots_impl(st, gt, nrows, m, nvec, basis, vstride, gsb, h, gsh, w, gsw, partial, gsp, out,
                        gso);
}
void launch_cols_update_dots_b(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                               const _Float16* basis, size_t vstride, size_t gsb, const double* h,
                               size_t gsh, double* w, size_t gsw, double* partial, size_t gsp,
                               double* out, size_t gso) {
  cols_update_dots_impl(st, gt, nrows, m, nvec, basis, vstride, gsb, h, gsh, w, gsw, partial, gsp, out,
                        gso);
}

// out[r,c] = scale[c] * ( w[r,c] + sign * sum_{i<nvec} h[i*m+c] * V_i[r,c] )
// (scale may be NULL = 1; w may be NULL = 0).  Streams nvec panels once.
// outf (optional): FP32 copy of the result (the stored Krylov vector); out then
// holds the same rounded values, so the vector the next operator application
// sees IS the stored one.
template <class BT>
__global__ __launch_bounds__(256) void cols_update_kernel(
    GroupTab gt, size_t nelem, int m, GroupInts nvecs, const BT* __restrict__ basis, size_t vstride,
    size_t gsb, const double* __restrict__ h, size_t gsh, double sign,
    const double* __restrict__ w, size_t gsw, const double* __restrict__ scale,
    double* __restrict__ out, size_t gso, BT* __restrict__ outf, size_t gsf) {
  const int grp = gt.gid[blockIdx.z];
  const int nvec = nvecs.v[grp];
  basis += (size_t)grp * gsb;
  h += (size_t)grp * gsh;
  if (w) w += (size_t)grp * gsw;
  if (scale) scale += (size_t)grp * m;
  if (out) out += (size_t)grp * gso;
  if (outf) outf += (size_t)grp * gsf;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < nelem;
       e += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % m);
    double s0 = 0.0, s1 = 0.0;
    int i = 0;
    for (; i + 1 < nvec; i += 2) {
      s0 = fma(h[i * m + c], (double)basis[(size_t)i * vstride + e], s0);
      s1 = fma(h[(i + 1) * m + c], (double)basis[(size_t)(i + 1) * vstride + e], s1);
    }
    if (i < nvec) s0 = fma(h[i * m + c], (double)basis[(size_t)i * vstride + e], s0);
    double v = (w ? w[e] : 0.0) + sign * (s0 + s1);
    if (scale) v *= scale[c];
    if (outf) {
      const BT f = (BT)v;
      outf[e] = f;
      v = (double)f;
    }
    if (out) out[e] = v;
  }
}
template <class BT>
static void cols_update_impl(hipStream_t st, const GroupTab& gt, int nrows, int m,
                             const GroupInts& nvec, const BT* basis, size_t vstride, size_t gsb,
                             const double* h, size_t gsh,
                             double sign, const double* w, size_t gsw, const double* scale,
                             double* out, size_t gso, BT* outf, size_t gsf) {
  size_t nelem = (size_t)nrows * m;
  if (!nelem || gt.ng <= 0) return;
  if constexpr (std::is_same<BT, _Float16>::value) {
    int nmax = 0;
    for (int i = 0; i < gt.ng; ++i) nmax = std::max(nmax, nvec.v[gt.gid[i]]);
    if ((m == 16 || ((m & 7) == 0 && m <= 32 && arnoldi16(8))) && arnoldi16(4) &&
        (size_t)nmax * m * sizeof(double) <= 48 * 1024) {
      const size_t nhalf = (size_t)nrows * (m / 8);       // 8-column pieces
      const int grid16 = (int)std::min<size_t>((nhalf + 255) / 256, 8192);
      hipLaunchKernelGGL(cols_update16_kernel, dim3(grid16, 1, gt.ng), dim3(256),
                         (size_t)std::max(nmax, 1) * m * sizeof(double), st, gt, nhalf, nvec, basis, vstride, gsb,
                         h, gsh, sign, w, gsw, scale, out, gso, outf, gsf, m);
      return;
    }
  }
  if constexpr (std::is_same<BT, float>::value) {
    int nmax = 0;
    for (int i = 0; i < gt.ng; ++i) nmax = std::max(nmax, nvec.v[gt.gid[i]]);
    if ((m & 3) == 0 && !outf && out && arnoldi16(4) && (size_t)nmax * m * sizeof(double) <= 48 * 1024) {
      const size_t nquad = nelem / 4;
      const int gridq = (int)std::min<size_t>((nquad + 255) / 256, 8192);
      hipLaunchKernelGGL(cols_update_f4_kernel, dim3(gridq, 1, gt.ng), dim3(256),
                         (size_t)std::max(nmax, 1) * m * sizeof(double), st, gt, nquad, nvec, basis, vstride, gsb, h,
                         gsh, sign, w, gsw, scale, out, gso, m);
      return;
    }
  }
  int grid = (int)std::min<size_t>((nelem + 255) / 256, 8192);
  hipLaunchKernelGGL(cols_update_kernel<BT>, dim3(grid, 1, gt.ng), dim3(256), 0, st, gt, nelem, m,
                     nvec, basis, vstride, gsb, h, gsh, sign, w, gsw, scale, out, gso, outf, gsf);
}
void launch_cols_update_b(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                          const double* basis, size_t vstride, size_t gsb, const double* h,
                          size_t gsh, double sign, const double* w, size_t gsw, const double* scale,
                          double* out, size_t gso) {
  cols_update_impl(st, gt, nrows, m, same_int(nvec), basis, vstride, gsb, h, gsh, sign, w, gsw, scale,
                   out, gso, (double*)nullptr, 0);
}
void launch_cols_update_b(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                          const float* basis, size_t vstride, size_t gsb, const double* h,
                          size_t gsh, double sign, const double* w, size_t gsw, const double* scale,
                          double* out, size_t gso, float* outf, size_t gsf) {
  cols_update_impl(st, gt, nrows, m, same_int(nvec), basis, vstride, gsb, h, gsh, sign, w, gsw, scale,
                   out, gso, outf, gsf);
}
void launch_cols_update_b(hipStream_t st, const GroupTab& gt, int nrows, int m, int nvec,
                          const _Float16* basis, size_t vstride, size_t gsb, const double* h,
                          size_t gsh, double sign, const double* w, size_t gsw, const double* scale,
                          double* out, size_t gso, _Float16* outf, size_t gsf) {
  cols_update_impl(st, gt, nrows, m, same_int(nvec), basis, vstride, gsb, h, gsh, sign, w, gsw, scale,
                   out, gso, outf, gsf);
}
// correction step of a restart cycle: group g combines its first nvec.v[g] vectors
void launch_cols_update_bk(hipStream_t st, const GroupTab& gt, int nrows, int m, const GroupInts& nvec,
                           const double* basis, size_t vstride, size_t gsb, const double* h, size_t gsh,
                           double* out, size_t gso, const double* acc, size_t gsa) {
  // acc (optional): out = acc + sum; acc may be `out` itself (every thread reads its elements before it writes them)
  cols_update_impl(st, gt, nrows, m, nvec, basis, vstride, gsb, h, gsh, 1.0, acc, gsa,
                   (const double*)nullptr, out, gso, (double*)nullptr, 0);
}
void launch_cols_update_bk(hipStream_t st, const GroupTab& gt, int nrows, int m, const GroupInts& nvec,
                           const _Float16* basis, size_t vstride, size_t gsb, const double* h, size_t gsh,
                           double* out, size_t gso, const double* acc, size_t gsa) {
  // acc (optional): out = acc + sum; acc may be `out` itself (every thread reads its elements before it writes them)
  cols_update_impl(st, gt, nrows, m, nvec, basis, vstride, gsb, h, gsh, 1.0, acc, gsa,
                   (const double*)nullptr, out, gso, (_Float16*)nullptr, 0);
}
void launch_cols_update_bk(hipStream_t st, const GroupTab& gt, int nrows, int m, const GroupInts& nvec,
                           const float* basis, size_t vstride, size_t gsb, const double* h, size_t gsh,
                           double* out, size_t gso, const double* acc, size_t gsa) {
  // acc (optional): out = acc + sum; acc may be `out` itself (every thread reads its elements before it writes them)
  cols_update_impl(st, gt, nrows, m, nvec, basis, vstride, gsb, h, gsh, 1.0, acc, gsa,
                   (const double*)nullptr, out, gso, (float*)nullptr, 0);
}
void launch_cols_update(hipStream_t st, int nrows, int m, int nvec, const double* basis,
                        size_t vstride, const double* h, double sign, const double* w,
                        const double* scale, double* out) {
  launch_cols_update_b(st, single_group(), nrows, m, nvec, basis, vstride, 0, h, 0, sign, w, 0, scale,
                       out, 0);
}

// ---------------------------------------------------------------------------
// GMRES small per-column kernels (one thread per panel column).
//   state layout (all device, per column c):
//     H   [c][j][i]   (restart+1) x restart upper Hessenberg -> R after rotations
//     cs,sn [c][i],  g [c][i]
// hess_update: consumes h1 (pass 1), h2 (pass 2 incl. ||w'||^2 as last row),
// applies the stored rotations, creates the new one, writes scale = 1/h_{j+1,j}
// (0 on breakdown / frozen column) and the residual estimate |g_{j+1}|.
// ---------------------------------------------------------------------------
// One 64-lane workgroup per panel column: the lanes stage h1+h2, cs, sn in LDS
// with independent loads (and reduce ||h2||^2 with shuffles); lane 0 then runs
// the sequential rotation chain out of LDS instead of a chain of dependent
// global loads.
__global__ __launch_bounds__(64) void gmres_hess_kernel(
    GroupTab gt, int m, int j, int restart, const double* __restrict__ h1,
    const double* __restrict__ h2, double* __restrict__ H, double* __restrict__ cs,
    double* __restrict__ sn, double* __restrict__ g, double* __restrict__ scale,
    double* __restrict__ resid, const double* __restrict__ bnorm, double tol,
    double* __restrict__ host_resid, double* __restrict__ zero_h1, double* __restrict__ zero_h2,
    double* __restrict__ hsum) {
  extern __shared__ double sh[];       // hcol[restart+2], csl[restart], snl[restart]
  if (host_resid) host_resid += (size_t)gt.gid[blockIdx.z] * m;
  {
    // group-major state: every array holds one slab per group
    const size_t grp = (size_t)gt.gid[blockIdx.z];
    h1 += grp * (restart + 2) * m;
    h2 += grp * (restart + 2) * m;
    if (zero_h1) zero_h1 += grp * (restart + 2) * m;
    if (hsum) hsum += grp * (restart + 2) * m;
    if (zero_h2) zero_h2 += grp * (restart + 2) * m;
    H += grp * m * (restart + 1) * restart;
    cs += grp * m * restart;
    sn += grp * m * restart;
    g += grp * m * (restart + 1);
    scale += grp * m;
    resid += grp * m;
    bnorm += grp * m;
  }
  double* hcol = sh;
  double* csl = sh + restart + 2;
  double* snl = csl + restart;
  const int c = blockIdx.x;
  const int lane = threadIdx.x;
  double* Hc = H + (size_t)c * (restart + 1) * restart + (size_t)j * (restart + 1);
  double* csc = cs + (size_t)c * restart;
  double* snc = sn + (size_t)c * restart;
  double* gc = g + (size_t)c * (restart + 1);
  const int nv = j + 1;
  double part = 0.0;
  for (int i = lane; i < nv; i += 64) {
    const double b = h2[i * m + c];
    part += b * b;
    hcol[i] = h1[i * m + c] + b;
    if (hsum) hsum[i * m + c] = hcol[i];       // coefficients of BOTH passes, for an update that starts from the unprojected w
  }
  // atomic dot passes (launch_cols_dots16_atomic): clear what has been consumed -- this column of the first-pass
  // sums, and of the second-pass buffer of the NEXT iteration (last read by the update of the previous one)
  if (zero_h1)
    for (int i = lane; i < nv; i += 64) zero_h1[i * m + c] = 0.0;
  if (zero_h2)
    for (int i = lane; i <= nv + 1 && i < restart + 2; i += 64) zero_h2[i * m + c] = 0.0;
  for (int i = lane; i < j; i += 64) {
    csl[i] = csc[i];
    snl[i] = snc[i];
  }
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
  __syncthreads();
  if (lane != 0) return;
  const double h2sq = part;
  const double ww = h2[nv * m + c];  // ||w'||^2 before the second projection
  const double gj = gc[j];
  double hn2 = ww - h2sq;
  double hnext = hn2 > 0.0 ? sqrt(hn2) : 0.0;
  // frozen column (already converged, or exact breakdown): keep it inert
  const double tiny = 1e-300;
  const bool dead = !(hnext > tiny) || (fabs(gj) <= 0.01 * tol * bnorm[c]);
  if (dead) hnext = 0.0;
  double cur = hcol[0];
  for (int i = 0; i < j; ++i) {
    const double nxt = hcol[i + 1];
    const double t = csl[i] * cur + snl[i] * nxt;
    const double u = -snl[i] * cur + csl[i] * nxt;
    Hc[i] = t;
    cur = u;
  }
  const double d = hypot(cur, hnext);
  double cj = 1.0, sj = 0.0;
  if (d > tiny) { cj = cur / d; sj = hnext / d; }
  csc[j] = cj;
  snc[j] = sj;
  Hc[j] = (d > tiny) ? d : 1.0;  // keep R non-singular for frozen columns
  Hc[j + 1] = 0.0;
  if (d > tiny) {
    gc[j + 1] = -sj * gj;
    gc[j] = cj * gj;
  } else {
    gc[j + 1] = 0.0;
    gc[j] = 0.0;
  }
  scale[c] = (hnext > tiny) ? 1.0 / hnext : 0.0;
  const double rnew = (d > tiny) ? fabs(sj * gj) : 0.0;
  resid[c] = rnew;
  // pinned host copy for the (lagged) convergence check: saves a D2H copy per iteration
  if (host_resid) host_resid[c] = rnew;
}
void launch_gmres_hess_b(hipStream_t st, const GroupTab& gt, int m, int j, int restart,
                         const double* h1, const double* h2, double* H, double* cs, double* sn,
                         double* g, double* scale, double* resid, const double* bnorm, double tol,
                         double* host_resid, double* zero_h1, double* zero_h2, double* hsum) {
  if (gt.ng <= 0) return;
  hipLaunchKernelGGL(gmres_hess_kernel, dim3(m, 1, gt.ng), dim3(64),
                     (3 * restart + 4) * sizeof(double), st, gt, m, j, restart, h1, h2, H, cs, sn, g,
                     scale, resid, bnorm, tol, host_resid, zero_h1, zero_h2, hsum);
}

// y[i*m + c] solves R y = g for the k x k triangle of column c.  One wave per (column, group): lane l first
// fetches column-entries R[i][l] = Hc[l][i] of all rows i <= l (independent loads, all in flight), then the k steps of
// the back substitution run out of LDS with a wave reduction each (one thread per column walking the triangle with
// dependent global loads took 31 us per call).
__global__ __launch_bounds__(64) void gmres_backsolve_kernel(GroupTab gt, int m, GroupInts ks, int restart,
                                                             const double* __restrict__ H,
                                                             const double* __restrict__ g,
                                                             double* __restrict__ y) {
  extern __shared__ double sm[];          // k rows of 64: sm[i * 64 + l] = R[i][l];  then ys[64]
  const int c = blockIdx.x, lane = threadIdx.x;
  const int k = ks.v[gt.gid[blockIdx.z]];
  if (k <= 0) return;
  {
    const size_t grp = (size_t)gt.gid[blockIdx.z];
    H += grp * m * (restart + 1) * restart;
    g += grp * m * (restart + 1);
    y += grp * restart * m;
  }
  const double* Hc = H + (size_t)c * (restart + 1) * restart;
  const double* gc = g + (size_t)c * (restart + 1);
  double* ys = sm + (size_t)k * 64;
  for (int i = 0; i < k; ++i)
    sm[i * 64 + lane] = (lane < k && lane >= i) ? Hc[(size_t)lane * (restart + 1) + i] : 0.0;
  ys[lane] = 0.0;
  __syncthreads();
  for (int i = k - 1; i >= 0; --i) {
    double part = (lane > i && lane < k) ? sm[i * 64 + lane] * ys[lane] : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if (lane == i) ys[i] = (gc[i] - part) / sm[i * 64 + i];
    __syncthreads();
  }
  if (lane < k) y[lane * m + c] = ys[lane];
}
// the same, one thread per column (cycles longer than a wave: gmres_restart > 63)
__global__ void gmres_backsolve_seq_kernel(GroupTab gt, int m, GroupInts ks, int restart,
                                           const double* __restrict__ H, const double* __restrict__ g,
                                           double* __restrict__ y) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= m) return;
  const int k = ks.v[gt.gid[blockIdx.z]];
  {
    const size_t grp = (size_t)gt.gid[blockIdx.z];
    H += grp * m * (restart + 1) * restart;
    g += grp * m * (restart + 1);
    y += grp * restart * m;
  }
  const double* Hc = H + (size_t)c * (restart + 1) * restart;
  const double* gc = g + (size_t)c * (restart + 1);
  for (int i = k - 1; i >= 0; --i) {
    double s = gc[i];
    for (int l = i + 1; l < k; ++l) s -= Hc[(size_t)l * (restart + 1) + i] * y[l * m + c];
    y[i * m + c] = s / Hc[(size_t)i * (restart + 1) + i];
  }
}
void launch_gmres_backsolve_b(hipStream_t st, const GroupTab& gt, int m, const GroupInts& k,
                              int restart, const double* H, const double* g, double* y) {
  if (gt.ng <= 0) return;
  if (restart > 63) {   // (the wave form holds one row per lane)
    hipLaunchKernelGGL(gmres_backsolve_seq_kernel, dim3((m + 63) / 64, 1, gt.ng), dim3(64), 0, st, gt, m, k, restart,
                       H, g, y);
    return;
  }
  hipLaunchKernelGGL(gmres_backsolve_kernel, dim3(m, 1, gt.ng), dim3(64), (size_t)(restart + 1) * 64 * sizeof(double),
                     st, gt, m, k, restart, H, g, y);
}

// start of a cycle: beta[c] = sqrt(nrm2[c]); g = [beta, 0...]; scale = 1/beta
__global__ void gmres_start_kernel(GroupTab gt, int m, int restart,
                                   const double* __restrict__ nrm2, double* __restrict__ g,
                                   double* __restrict__ scale, double* __restrict__ resid) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= m) return;
  {
    const size_t grp = (size_t)gt.gid[blockIdx.z];
    nrm2 += grp * m;
    g += grp * m * (restart + 1);
    scale += grp * m;
    resid += grp * m;
  }
  const double b = sqrt(fmax(nrm2[c], 0.0));
  double* gc = g + (size_t)c * (restart + 1);
  for (int i = 0; i <= restart; ++i) gc[i] = 0.0;
  gc[0] = b;
  scale[c] = b > 1e-300 ? 1.0 / b : 0.0;
  resid[c] = b;
}
void launch_gmres_start_b(hipStream_t st, const GroupTab& gt, int m, int restart,
                          const double* nrm2, double* g, double* scale, double* resid) {
  if (gt.ng <= 0) return;
  hipLaunchKernelGGL(gmres_start_kernel, dim3((m + 63) / 64, 1, gt.ng), dim3(64), 0, st, gt, m,
                     restart, nrm2, g, scale, resid);
}

// ---------------------------------------------------------------------------
// K2: block-Jacobi.  Blocks are BS x BS dense inverses (padded with identity),
// members listed in `rows`.  One wave per block: lane (s, c) = (lane>>4,
// lane&15) holds the block's input column c in registers and produces the
// output rows s, s+4, ...
//   out[rows[il], :] = sum_jl inv[b][il][jl] * in[rows[jl], :]
// ---------------------------------------------------------------------------
// four consecutive entries of a stored inverse (FP64: two 16-B loads, FP32: one)
__device__ __forceinline__ void load4(const double* p, double (&a)[4]) {
  const double2 u = reinterpret_cast<const double2*>(p)[0], v = reinterpret_cast<const double2*>(p)[1];
  a[0] = u.x; a[1] = u.y; a[2] = v.x; a[3] = v.y;
}
__device__ __forceinline__ void load4(const float* p, double (&a)[4]) {
  const float4 u = reinterpret_cast<const float4*>(p)[0];
  a[0] = (double)u.x; a[1] = (double)u.y; a[2] = (double)u.z; a[3] = (double)u.w;
}

// four consecutive stored entries as ONE raw load; converted to FP64 only when used (a conversion between
// loads makes the compiler wait for each load in turn)
template <class T>
struct Raw4;
template <>
struct Raw4<float> {
  typedef float4 type;
  static __device__ __forceinline__ float4 load(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void unpack(const float4& u, double (&a)[4]) {
    a[0] = (double)u.x; a[1] = (double)u.y; a[2] = (double)u.z; a[3] = (double)u.w;
  }
};
template <>
struct Raw4<double> {
  struct type { double2 lo, hi; };
  static __device__ __forceinline__ type load(const double* p) {
    type t;
    t.lo = reinterpret_cast<const double2*>(p)[0];
    t.hi = reinterpret_cast<const double2*>(p)[1];
    return t;
  }
  static __device__ __forceinline__ void unpack(const type& u, double (&a)[4]) {
    a[0] = u.lo.x; a[1] = u.lo.y; a[2] = u.hi.x; a[3] = u.hi.y;
  }
};
template <int BS, class T>
__global__ __launch_bounds__(256) void block_apply_kernel(
    GroupTab gt, int nblocks, const int* __restrict__ bptr, const int* __restrict__ rows,
    GroupPtrsT<T> invs, const double* __restrict__ in, int ldi, size_t gsi,
    double* __restrict__ out, int ldo, size_t gso, int m, int subtract, ProlongArgs pa,
    CsrInArgs ci) {
  const int grp = gt.gid[blockIdx.z];
  const T* __restrict__ inv = invs.p[grp];
  in += (size_t)grp * gsi;
  out += (size_t)grp * gso;
  const double* __restrict__ csrc = ci.rp ? ci.src + (size_t)grp * ci.gss : nullptr;
  const double* __restrict__ cval = ci.v.p[grp];
  const double* __restrict__ cbase = ci.base ? ci.base + (size_t)grp * ci.gsb : nullptr;
  const double* __restrict__ ec = pa.aggof ? pa.ec + (size_t)grp * pa.gse : nullptr;
  // One wave per block, FP64 MFMA 16x16x4: out_tile (16 rows x 16 cols) +=
  // inv[rows 16*ti.., k] * x[k, cols].  A-operand lane (r = l&15, q = l>>4)
  // holds inv[16*ti + r][k0 + 4q + s] for MFMA s of a 16-wide k chunk (one
  // 32-B load per lane and chunk); the matching B operand is the gathered
  // input row rows[k0 + 4q + s], column c0 + r.
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  if (wave >= nblocks) {
    // surplus waves: coarse-level prolongation of the rows outside the blocks
    // (the pressure rows when this is the last velocity sweep), 32 rows per wave
    const int e0 = (wave - nblocks) * 32;
    for (int rr = e0 + q; rr < min(e0 + 32, pa.nextra); rr += 4) {
      const int row = pa.row0 + rr;
      for (int col = r; col < m; col += 16) {
        const double v = out[(size_t)row * ldo + col] + ec[(size_t)pa.aggof[row] * m + col];
        out[(size_t)row * ldo + col] = v;
        if (pa.out32) pa.out32[(size_t)grp * pa.gs32 + (size_t)row * ldo + col] = (float)v;
      }
    }
    return;
  }
  const int b0 = bptr[wave], nb = bptr[wave + 1] - b0;
  const T* Bi = inv + (size_t)wave * BS * BS;
  constexpr int NT = BS / 16;
  for (int c0 = 0; c0 < m; c0 += 16) {
    const int col = c0 + r;
    d4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kc = 0; kc < NT; ++kc) {          // 16-wide chunks of the block's columns
      double xb[4];
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const int kk = kc * 16 + 4 * q + s2;
        if (!csrc) {
          xb[s2] = (kk < nb && col < m) ? in[(size_t)rows[b0 + kk] * ldi + col] : 0.0;
        } else {
          // input row computed on the fly: base[row] + scale * (C * src)[row] with the CSR
          // matrix C (J^T product of the SIMPLE sweep / residual after the coarse
          // correction; src is small and L2 resident)
          double acc0 = 0.0, acc1 = 0.0, bv = 0.0;
          if (kk < nb && col < m) {
            const int row = rows[b0 + kk];
            int k = ci.rp[row];
            const int k1 = ci.rp[row + 1];
            if (cbase) bv = cbase[(size_t)row * ldi + col];
            for (; k + 1 < k1; k += 2) {
              acc0 = fma(cval[k], csrc[(size_t)ci.ci[k] * ldi + col], acc0);
              acc1 = fma(cval[k + 1], csrc[(size_t)ci.ci[k + 1] * ldi + col], acc1);
            }
            if (k < k1) acc0 = fma(cval[k], csrc[(size_t)ci.ci[k] * ldi + col], acc0);
          }
          xb[s2] = fma(ci.scale, acc0 + acc1, bv);
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        double a4[4];
        load4(Bi + (size_t)(16 * t + r) * BS + kc * 16 + 4 * q, a4);
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[0], xb[0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[1], xb[1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[2], xb[2], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[3], xb[3], acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int il = 16 * t + q + 4 * e;
        if (il < nb && col < m) {
          const int row = rows[b0 + il];
          double* o = &out[(size_t)row * ldo + col];
          double v = subtract ? *o - acc[t][e] : acc[t][e];
          // optional second copy WITHOUT the coarse part (group stride pa.gs2)
          if (pa.out2) pa.out2[(size_t)grp * pa.gs2 + (size_t)row * ldo + col] = v;
          if (ec) v += ec[(size_t)pa.aggof[row] * m + col];   // fused coarse-level prolongation
          if (!(pa.out32 && pa.only32)) *o = v;
          if (pa.out32) pa.out32[(size_t)grp * pa.gs32 + (size_t)row * ldo + col] = (float)v;
        }
      }
  }
}
template <class T>
static void block_apply_rect_impl(hipStream_t st, const GroupTab& gt, int bs, int ks, int nblocks,
                                  const int* bptr, const int* rows, const int* iptr, const int* irows,
                                  const GroupPtrsT<T>& mats, const double* in, int ldi, size_t gsi, double* out,
                                  int ldo, size_t gso, int m, int subtract, const ProlongArgs& pa);
template <class T>
static void block_apply_impl(hipStream_t st, const GroupTab& gt, int bs, int nblocks, const int* bptr,
                             const int* rows, const GroupPtrsT<T>& inv, const double* in, int ldi,
                             size_t gsi, double* out, int ldo, size_t gso, int m, int subtract,
                             const ProlongArgs& pa, const CsrInArgs& ci) {
  if (nblocks <= 0 || gt.ng <= 0) return;
  // plain panel input, 32 x 32 blocks: the rectangle kernel with the block's own rows as its input list (its loads
  // are issued in groups; this kernel's index -> gather pairs are a chain of dependent round trips).
  // RICADI_BA_PLAIN=1 keeps this kernel.
  // Only where the launch is latency bound (few waves: the Schur sweep of cfg2 has 110 blocks x 16 groups): with many
  // waves the rectangle kernel's 152 VGPRs cost more than its grouped loads gain (velocity-sized sweep at cfg2:
  // 43 vs 33 us).
  static const bool via_rect = true;
  if (via_rect && !ci.rp && bs == 32 && (long)nblocks * gt.ng <= 8192) {
    block_apply_rect_impl(st, gt, 32, 32, nblocks, bptr, rows, bptr, rows, inv, in, ldi, gsi, out, ldo, gso, m,
                          subtract, pa);
    return;
  }
  const int nwaves = nblocks + (pa.aggof ? (pa.nextra + 31) / 32 : 0);
  dim3 grid((nwaves + 3) / 4, 1, gt.ng), block(256);
  switch (bs) {
    case 16:
      hipLaunchKernelGGL((block_apply_kernel<16, T>), grid, block, 0, st, gt, nblocks, bptr, rows,
                         inv, in, ldi, gsi, out, ldo, gso, m, subtract, pa, ci);
      break;
    case 32:
      hipLaunchKernelGGL((block_apply_kernel<32, T>), grid, block, 0, st, gt, nblocks, bptr, rows,
                         inv, in, ldi, gsi, out, ldo, gso, m, subtract, pa, ci);
      break;
    default:
      hipLaunchKernelGGL((block_apply_kernel<64, T>), grid, block, 0, st, gt, nblocks, bptr, rows,
                         inv, in, ldi, gsi, out, ldo, gso, m, subtract, pa, ci);
      break;
  }
}
void launch_block_apply_b(hipStream_t st, const GroupTab& gt, int bs, int nblocks, const int* bptr,
                          const int* rows, const GroupPtrs& inv, const double* in, int ldi,
                          size_t gsi, double* out, int ldo, size_t gso, int m, int subtract,
                          const ProlongArgs& pa, const CsrInArgs& ci) {
  block_apply_impl(st, gt, bs, nblocks, bptr, rows, inv, in, ldi, gsi, out, ldo, gso, m, subtract, pa,
                   ci);
}
void launch_block_apply_b(hipStream_t st, const GroupTab& gt, int bs, int nblocks, const int* bptr,
                          const int* rows, const GroupPtrsF& inv, const double* in, int ldi,
                          size_t gsi, double* out, int ldo, size_t gso, int m, int subtract,
                          const ProlongArgs& pa, const CsrInArgs& ci) {
  block_apply_impl(st, gt, bs, nblocks, bptr, rows, inv, in, ldi, gsi, out, ldo, gso, m, subtract, pa,
                   ci);
}

// ---------------------------------------------------------------------------
// K2, rectangular form: the last velocity sweep of the SIMPLE cycle,
//     z_v[rows_b] -= G_b * z_p[pcols_b],      G_b = Ahat_b^-1 * J^T[rows_b, pcols_b]   (BS x KS)
// with the per-shift product G_b formed once at setup (gt_blocks_kernel) from the block-Jacobi
// inverse and the dense slice of J^T over the block's rows and the pressure dofs they touch.
// Replaces the same sweep with the J^T rows gathered entry by entry inside the kernel (CsrInArgs:
// ~17 dependent loads per operand element, 89 us per 16-group launch at cfg2 against 34 us for
// a plain sweep).  One wave per block, FP64 MFMA 16x16x4 as in block_apply_kernel; the input
// rows come from their own list (pressure-local indices), the output rows from the block's.
// ---------------------------------------------------------------------------
template <int BS, int KS, class T>
__global__ __launch_bounds__(256) void block_apply_rect_kernel(
    GroupTab gt, int nblocks, const int* __restrict__ bptr, const int* __restrict__ rows,
    const int* __restrict__ iptr, const int* __restrict__ irows, GroupPtrsT<T> mats,
    const double* __restrict__ in, int ldi, size_t gsi, double* __restrict__ out, int ldo,
    size_t gso, int m, int subtract, ProlongArgs pa) {
  const int grp = gt.gid[blockIdx.z];
  const T* __restrict__ mat = mats.p[grp];
  in += (size_t)grp * gsi;
  out += (size_t)grp * gso;
  const double* __restrict__ ec = pa.aggof ? pa.ec + (size_t)grp * pa.gse : nullptr;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  if (wave >= nblocks) {
    // surplus waves: coarse-level prolongation of the rows outside the blocks
    const int e0 = (wave - nblocks) * 32;
    for (int rr = e0 + q; rr < min(e0 + 32, pa.nextra); rr += 4) {
      const int row = pa.row0 + rr;
      for (int col = r; col < m; col += 16) {
        const double v = out[(size_t)row * ldo + col] + ec[(size_t)pa.aggof[row] * m + col];
        out[(size_t)row * ldo + col] = v;
        if (pa.out32) pa.out32[(size_t)grp * pa.gs32 + (size_t)row * ldo + col] = (float)v;
      }
    }
    return;
  }
  const int b0 = bptr[wave], nb = bptr[wave + 1] - b0;
  const int i0 = iptr[wave], ni = iptr[wave + 1] - i0;
  const T* Gi = mat + (size_t)wave * BS * KS;
  constexpr int NT = BS / 16, NK = KS / 16;
  // Every index this wave needs is loaded up front and WITHOUT conditions (clamped to a valid entry, masked
  // at use): an index load inside the condition of its gather made the compiler wait for every pair in turn
  // -- the kernel was a chain of ~20 dependent round trips.
  int xrow[NK][4], orow[NT][4], oagg[NT][4];
#pragma unroll
  for (int kc = 0; kc < NK; ++kc)
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      const int kk = kc * 16 + 4 * q + s2;
      xrow[kc][s2] = ni > 0 ? irows[i0 + min(kk, ni - 1)] : 0;
    }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) orow[t][e] = rows[b0 + min(16 * t + q + 4 * e, nb - 1)];
  const int* __restrict__ aggp = ec ? pa.aggof : rows;        // a readable dummy when there is no coarse part
  const double* __restrict__ ecp = ec ? ec : out;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) oagg[t][e] = aggp[orow[t][e]];
  for (int c0 = 0; c0 < m; c0 += 16) {
    const int col = c0 + r;
    const bool cok = col < m;
    const int colx = cok ? col : 0;
    // all gathers, all tile loads, then (old output, coarse part) -- each group issued together
    double xb[NK][4];
    typename Raw4<T>::type graw[NT][NK];
#pragma unroll
    for (int kc = 0; kc < NK; ++kc)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) xb[kc][s2] = in[(size_t)xrow[kc][s2] * ldi + colx];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int kc = 0; kc < NK; ++kc) graw[t][kc] = Raw4<T>::load(Gi + (size_t)(16 * t + r) * KS + kc * 16 + 4 * q);
    __builtin_amdgcn_sched_barrier(0);
    d4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kc = 0; kc < NK; ++kc) {
      if (kc * 16 >= ni) break;                 // wave-uniform: chunks beyond the block's inputs
      double xm[4];
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) xm[s2] = (kc * 16 + 4 * q + s2 < ni && cok) ? xb[kc][s2] : 0.0;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        double a4[4];
        Raw4<T>::unpack(graw[t][kc], a4);
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[0], xm[0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[1], xm[1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[2], xm[2], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[3], xm[3], acc[t], 0, 0, 0);
      }
    }
    // old output and coarse part: again all loads together, without conditions
    double oldv[NT][4], ecv[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        oldv[t][e] = out[(size_t)orow[t][e] * ldo + colx];
        ecv[t][e] = ecp[(size_t)oagg[t][e] * m + colx];
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int il = 16 * t + q + 4 * e;
        if (il < nb && cok) {
          const size_t at = (size_t)orow[t][e] * ldo + col;
          double v = subtract ? oldv[t][e] - acc[t][e] : acc[t][e];
          if (pa.out2) pa.out2[(size_t)grp * pa.gs2 + at] = v;        // the result before the coarse part
          if (ec) v += ecv[t][e];
          if (!(pa.out32 && pa.only32)) out[at] = v;
          if (pa.out32) pa.out32[(size_t)grp * pa.gs32 + at] = (float)v;
        }
      }
  }
}
template <class T>
static void block_apply_rect_impl(hipStream_t st, const GroupTab& gt, int bs, int ks, int nblocks,
                                  const int* bptr, const int* rows, const int* iptr, const int* irows,
                                  const GroupPtrsT<T>& mats, const double* in, int ldi, size_t gsi, double* out,
                                  int ldo, size_t gso, int m, int subtract, const ProlongArgs& pa) {
  if (nblocks <= 0 || gt.ng <= 0) return;
  const int nwaves = nblocks + (pa.aggof ? (pa.nextra + 31) / 32 : 0);
  dim3 grid((nwaves + 3) / 4, 1, gt.ng), block(256);
#define RICADI_RECT(B, K)                                                                           \
  hipLaunchKernelGGL((block_apply_rect_kernel<B, K, T>), grid, block, 0, st, gt, nblocks, bptr, rows, \
                     iptr, irows, mats, in, ldi, gsi, out, ldo, gso, m, subtract, pa)
  if (bs == 32 && ks == 32) RICADI_RECT(32, 32);
  else if (bs == 32 && ks == 64) RICADI_RECT(32, 64);
  else if (bs == 16 && ks == 32) RICADI_RECT(16, 32);
  else if (bs == 16 && ks == 64) RICADI_RECT(16, 64);
  else if (bs == 64 && ks == 64) RICADI_RECT(64, 64);
  else RICADI_RECT(64, 128);
#undef RICADI_RECT
}
bool block_apply_rect_ok(int bs, int ks) {
  return (bs == 32 && (ks == 32 || ks == 64)) || (bs == 16 && (ks == 32 || ks == 64)) ||
         (bs == 64 && (ks == 64 || ks == 128));
}
void launch_block_apply_rect_b(hipStream_t st, const GroupTab& gt, int bs, int ks, int nblocks,
                               const int* bptr, const int* rows, const int* iptr, const int* irows,
                               const GroupPtrs& mats, const double* in, int ldi, size_t gsi, double* out,
                               int ldo, size_t gso, int m, int subtract, const ProlongArgs& pa) {
  block_apply_rect_impl(st, gt, bs, ks, nblocks, bptr, rows, iptr, irows, mats, in, ldi, gsi, out, ldo, gso, m,
                        subtract, pa);
}
void launch_block_apply_rect_b(hipStream_t st, const GroupTab& gt, int bs, int ks, int nblocks,
                               const int* bptr, const int* rows, const int* iptr, const int* irows,
                               const GroupPtrsF& mats, const double* in, int ldi, size_t gsi, double* out,
                               int ldo, size_t gso, int m, int subtract, const ProlongArgs& pa) {
  block_apply_rect_impl(st, gt, bs, ks, nblocks, bptr, rows, iptr, irows, mats, in, ldi, gsi, out, ldo, gso, m,
                        subtract, pa);
}

// ---------------------------------------------------------------------------
// K2, two-term form:   out[rows_b] = M1_b * in1[list1_b] - M2_b * in2[list2_b]
// (+ prolongation / plain copy through ProlongArgs).  Used for the FIRST velocity sweep with the
// residual of the coarse correction folded in,
//     z_v = Ahat_b^-1 (r_v - (S Y e)_v)[rows_b] = Ahat_b^-1 r_v[rows_b] - (Ahat_b^-1 D_b) e[ccols_b],
// D_b = the dense slice of the prolongated operator S*Y over the block's rows and the <= 32
// coarse columns they touch, Ahat_b^-1 D_b formed per shift at setup (ady_blocks_kernel): the
// pass that wrote r - (S Y) e for all n rows and the re-read of it disappear (only the pressure
// rows still go through a small CSR product).  A segment with list == NULL takes the block's
// own rows.  One wave per block, FP64 MFMA 16x16x4 as in block_apply_kernel.
// ---------------------------------------------------------------------------
// Segment 1 is always the block's own rows with a BS x BS matrix; segment 2 has a compile-time
// padded width K2 (its list may be shorter).  All index loads are issued first, then all gathers,
// then the MFMAs: the two segments' dependent-load chains overlap instead of following each other.
template <int BS, int K2, class T, bool H1 = false>
__global__ __launch_bounds__(256) void block_apply2_kernel(
    GroupTab gt, int nblocks, const int* __restrict__ bptr, const int* __restrict__ rows,
    GroupPtrsT<T> m1s, Seg2 s1, GroupPtrsT<T> m2s, Seg2 s2, double* __restrict__ out, int ldo, size_t gso,
    int m, ProlongArgs pa) {
  const int grp = gt.gid[blockIdx.z];
  out += (size_t)grp * gso;
  const double* __restrict__ ec = pa.aggof ? pa.ec + (size_t)grp * pa.gse : nullptr;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  if (wave >= nblocks) {
    const int e0 = (wave - nblocks) * 32;
    for (int rr = e0 + q; rr < min(e0 + 32, pa.nextra); rr += 4) {
      const int row = pa.row0 + rr;
      for (int col = r; col < m; col += 16) {
        const double v = out[(size_t)row * ldo + col] + ec[(size_t)pa.aggof[row] * m + col];
        out[(size_t)row * ldo + col] = v;
        if (pa.out32) pa.out32[(size_t)grp * pa.gs32 + (size_t)row * ldo + col] = (float)v;
      }
    }
    return;
  }
  const int b0 = bptr[wave], nb = bptr[wave + 1] - b0;
  const int i0 = s2.iptr[wave], ni = s2.iptr[wave + 1] - i0;
  constexpr int NT = BS / 16, N1 = BS / 16, N2 = K2 / 16;
  const T* __restrict__ M1 = m1s.p[grp] + (size_t)wave * BS * BS;
  const T* __restrict__ M2 = m2s.p[grp] + (size_t)wave * BS * K2;
  const double* __restrict__ in1 = s1.in ? s1.in + (size_t)grp * s1.gs : nullptr;
  const _Float16* __restrict__ in1h = s1.in16 ? s1.in16 + (size_t)grp * s1.gs : nullptr;
  const double* __restrict__ in2 = s2.in + (size_t)grp * s2.gs;
  // input row ids of this lane: 4 per 16-wide chunk
  int r1[N1][4], r2[N2][4];
#pragma unroll
  for (int kc = 0; kc < N1; ++kc)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int kk = kc * 16 + 4 * q + s4;
      r1[kc][s4] = kk < nb ? rows[b0 + kk] : -1;
    }
#pragma unroll
  for (int kc = 0; kc < N2; ++kc)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int kk = kc * 16 + 4 * q + s4;
      r2[kc][s4] = kk < ni ? s2.irows[i0 + kk] : -1;
    }
  for (int c0 = 0; c0 < m; c0 += 16) {
    const int col = c0 + r;
    const bool cok = col < m;
    double x1[N1][4], x2[N2][4];
    if (H1) {
      // raw FP16 loads first, conversion afterwards: a conversion between the loads makes the compiler wait
      // for each of them in turn (the sweep was 37.8 instead of 32.1 us with the FP16 input)
      _Float16 h1[N1][4];
#pragma unroll
      for (int kc = 0; kc < N1; ++kc)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          h1[kc][s4] = in1h[(r1[kc][s4] >= 0 && cok) ? (size_t)r1[kc][s4] * m + col : (size_t)0];   // unconditional load
#pragma unroll
      for (int kc = 0; kc < N2; ++kc)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          x2[kc][s4] = (r2[kc][s4] >= 0 && cok) ? in2[(size_t)r2[kc][s4] * m + col] : 0.0;
#pragma unroll
      for (int kc = 0; kc < N1; ++kc)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) x1[kc][s4] = (r1[kc][s4] >= 0 && cok) ? (double)h1[kc][s4] : 0.0;
    } else {
#pragma unroll
    for (int kc = 0; kc < N1; ++kc)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
        x1[kc][s4] = (r1[kc][s4] >= 0 && cok) ? in1[(size_t)r1[kc][s4] * m + col] : 0.0;
#pragma unroll
    for (int kc = 0; kc < N2; ++kc)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
        x2[kc][s4] = (r2[kc][s4] >= 0 && cok) ? in2[(size_t)r2[kc][s4] * m + col] : 0.0;
    }
    d4 acc1[NT], acc2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc1[t] = acc2[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kc = 0; kc < N1; ++kc)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        double a4[4];
        load4(M1 + (size_t)(16 * t + r) * BS + kc * 16 + 4 * q, a4);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          acc1[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[s4], x1[kc][s4], acc1[t], 0, 0, 0);
      }
#pragma unroll
    for (int kc = 0; kc < N2; ++kc) {
      if (kc * 16 >= ni) break;                 // wave-uniform
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        double a4[4];
        load4(M2 + (size_t)(16 * t + r) * K2 + kc * 16 + 4 * q, a4);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          acc2[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[s4], x2[kc][s4], acc2[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int il = 16 * t + q + 4 * e;
        if (il < nb && cok) {
          const int row = rows[b0 + il];
          double v = acc1[t][e] - acc2[t][e];
          if (pa.out2) pa.out2[(size_t)grp * pa.gs2 + (size_t)row * ldo + col] = v;
          if (ec) v += ec[(size_t)pa.aggof[row] * m + col];
          out[(size_t)row * ldo + col] = v;
        }
      }
  }
}
bool block_apply2_ok(int bs, int k2) { return (bs == 32 || bs == 16) && (k2 == 32 || k2 == 64); }
template <class T>
static void block_apply2_impl(hipStream_t st, const GroupTab& gt, int bs, int nblocks, const int* bptr,
                              const int* rows, const GroupPtrsT<T>& m1, const Seg2& s1,
                              const GroupPtrsT<T>& m2, const Seg2& s2, double* out, int ldo, size_t gso, int m,
                              const ProlongArgs& pa) {
  if (nblocks <= 0 || gt.ng <= 0) return;
  const int nwaves = nblocks + (pa.aggof ? (pa.nextra + 31) / 32 : 0);
  dim3 grid((nwaves + 3) / 4, 1, gt.ng), block(256);
  // H = first-segment rows read from an FP16 panel (s1.in16): its own instantiation -- both paths in one
  // kernel cost 148 instead of 128 VGPRs, i.e. one wave per SIMD less
#define RICADI_BA2(B, K, H)                                                                              \
  hipLaunchKernelGGL((block_apply2_kernel<B, K, T, H>), grid, block, 0, st, gt, nblocks, bptr, rows, m1, s1, m2, \
                     s2, out, ldo, gso, m, pa)
  if (s1.in16) {
    if (bs == 32 && s2.kstride == 32) RICADI_BA2(32, 32, true);
    else if (bs == 32) RICADI_BA2(32, 64, true);
    else if (s2.kstride == 32) RICADI_BA2(16, 32, true);
    else RICADI_BA2(16, 64, true);
  } else {
    if (bs == 32 && s2.kstride == 32) RICADI_BA2(32, 32, false);
    else if (bs == 32) RICADI_BA2(32, 64, false);
    else if (s2.kstride == 32) RICADI_BA2(16, 32, false);
    else RICADI_BA2(16, 64, false);
  }
#undef RICADI_BA2
}
void launch_block_apply2_b(hipStream_t st, const GroupTab& gt, int bs, int nblocks, const int* bptr,
                           const int* rows, const GroupPtrs& m1, const Seg2& s1, const GroupPtrs& m2,
                           const Seg2& s2, double* out, int ldo, size_t gso, int m, const ProlongArgs& pa) {
  block_apply2_impl(st, gt, bs, nblocks, bptr, rows, m1, s1, m2, s2, out, ldo, gso, m, pa);
}
void launch_block_apply2_b(hipStream_t st, const GroupTab& gt, int bs, int nblocks, const int* bptr,
                           const int* rows, const GroupPtrsF& m1, const Seg2& s1, const GroupPtrsF& m2,
                           const Seg2& s2, double* out, int ldo, size_t gso, int m, const ProlongArgs& pa) {
  block_apply2_impl(st, gt, bs, nblocks, bptr, rows, m1, s1, m2, s2, out, ldo, gso, m, pa);
}

// out[b] = Ainv[b] (bs x bs) * (alpha_s dE[b] + beta_s dA[b] + dJ[b]) (bs x ks)  for every velocity block
// b and every shift s of the setup (blockIdx.y): the dense slices of the prolongated operator S*Y
// combined for the shift, times the block-Jacobi inverse.  One workgroup per block.
struct ShiftCoefs {
  double alpha[RICADI_MAX_GROUPS], beta[RICADI_MAX_GROUPS];
};
__global__ __launch_bounds__(256) void ady_blocks_kernel(int bs, int ks, const double* __restrict__ dA,
                                                         const double* __restrict__ dE,
                                                         const double* __restrict__ dJ,
                                                         const double* __restrict__ dT, ShiftCoefs cf,
                                                         GroupPtrs ainvs, GroupPtrs outs) {
  extern __shared__ double sm[];            // Ai (bs x bs), D (bs x ks)
  double* Ai = sm;
  double* D = sm + bs * bs;
  const double al = cf.alpha[blockIdx.y], be = cf.beta[blockIdx.y];
  const double* __restrict__ ainv = ainvs.p[blockIdx.y] + (size_t)blockIdx.x * bs * bs;
  const size_t off = (size_t)blockIdx.x * bs * ks;
  double* __restrict__ out = const_cast<double*>(outs.p[blockIdx.y]) + off;
  for (int e = threadIdx.x; e < bs * bs; e += 256) Ai[e] = ainv[e];
  for (int e = threadIdx.x; e < bs * ks; e += 256) D[e] = al * dE[off + e] + be * dA[off + e] + dJ[off + e];
  __syncthreads();
  for (int e = threadIdx.x; e < bs * ks; e += 256) {
    const int i = e / ks, j = e - i * ks;
    double sacc = 0.0;
    for (int t = 0; t < bs; ++t) sacc = fma(Ai[i * bs + t], D[t * ks + j], sacc);
    // smoothed aggregation: the sweep subtracts (out e); (P - Y) e is ADDED to the sweep's result there
    out[e] = dT ? sacc - dT[off + e] : sacc;
  }
}
void launch_ady_blocks(hipStream_t st, int nshift, int nblocks, int bs, int ks, const double* dA,
                       const double* dE, const double* dJ, const double* dT, const double* alphas,
                       const double* betas, const GroupPtrs& ainv, const GroupPtrs& out) {
  if (nblocks <= 0 || nshift <= 0) return;
  ShiftCoefs cf;
  for (int i = 0; i < RICADI_MAX_GROUPS; ++i) {
    cf.alpha[i] = i < nshift ? alphas[i] : 0.0;
    cf.beta[i] = i < nshift ? betas[i] : 0.0;
  }
  hipLaunchKernelGGL(ady_blocks_kernel, dim3(nblocks, nshift), dim3(256),
                     (size_t)(bs * bs + bs * ks) * sizeof(double), st, bs, ks, dA, dE, dJ, dT, cf, ainv, out);
}

// G[b] = Ainv[b] (bs x bs) * JTd[b] (bs x ks)  for every velocity block b and every shift of the
// setup (blockIdx.y); JTd is the dense slice of J^T (shift independent).  One workgroup per block.
__global__ __launch_bounds__(256) void gt_blocks_kernel(int bs, int ks, const double* __restrict__ jtd,
                                                        GroupPtrs ainvs, GroupPtrs outs) {
  extern __shared__ double sm[];            // Ai (bs x bs), Jd (bs x ks)
  double* Ai = sm;
  double* Jd = sm + bs * bs;
  const double* __restrict__ ainv = ainvs.p[blockIdx.y] + (size_t)blockIdx.x * bs * bs;
  const double* __restrict__ jsrc = jtd + (size_t)blockIdx.x * bs * ks;
  double* __restrict__ out = const_cast<double*>(outs.p[blockIdx.y]) + (size_t)blockIdx.x * bs * ks;
  for (int e = threadIdx.x; e < bs * bs; e += 256) Ai[e] = ainv[e];
  for (int e = threadIdx.x; e < bs * ks; e += 256) Jd[e] = jsrc[e];
  __syncthreads();
  for (int e = threadIdx.x; e < bs * ks; e += 256) {
    const int i = e / ks, j = e - i * ks;
    double sacc = 0.0;
    for (int t = 0; t < bs; ++t) sacc = fma(Ai[i * bs + t], Jd[t * ks + j], sacc);
    out[e] = sacc;
  }
}
void launch_gt_blocks(hipStream_t st, int nshift, int nblocks, int bs, int ks, const double* jtd,
                      const GroupPtrs& ainv, const GroupPtrs& out) {
  if (nblocks <= 0 || nshift <= 0) return;
  hipLaunchKernelGGL(gt_blocks_kernel, dim3(nblocks, nshift), dim3(256),
                     (size_t)(bs * bs + bs * ks) * sizeof(double), st, bs, ks, jtd, ainv, out);
}

// blocks[b] = alpha*Be[b] + beta*Ba[b]  (dense, bs x bs each)
__global__ void block_combine_kernel(size_t n, const double* Ba, const double* Be, double alpha,
                                     double beta, double* out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    out[i] = alpha * Be[i] + beta * Ba[i];
}
void launch_block_combine(hipStream_t st, size_t n, const double* Ba, const double* Be,
                          double alpha, double beta, double* out) {
  int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(block_combine_kernel, dim3(grid), dim3(256), 0, st, n, Ba, Be, alpha, beta,
                     out);
}


// Diagonal blocks of the CONSISTENT SIMPLE Schur complement
//   S_bb = sum_beta J_{b,beta} * Ahat_beta^-1 * J_{b,beta}^T
// (Ahat^-1 = the block-Jacobi inverse actually applied to the velocity block,
// not its diagonal).  One workgroup per pressure block loops over the coupled
// velocity blocks; both bs x bs products go through LDS.  Measured on the CPU
// mirror (N = 58): GMRES iterations 170 / 115 / 61 -> 108 / 74 / 43.
__global__ __launch_bounds__(256) void schur_blocks_bj_kernel(
    int bs, const int* __restrict__ bptr, const int* __restrict__ jd_ptr,
    const int* __restrict__ jd_vblk, const double* __restrict__ jd_val,
    GroupPtrs bvinvs, GroupPtrs blockss) {
  // one launch serves all shifts being set up: blockIdx.y = shift
  const double* __restrict__ bvinv = bvinvs.p[blockIdx.y];
  double* __restrict__ blocks = const_cast<double*>(blockss.p[blockIdx.y]);
  extern __shared__ double sm[];           // Jd, Ai, T : 3 x bs x bs
  double* Jd = sm;
  double* Ai = sm + bs * bs;
  double* T = sm + 2 * bs * bs;
  const int b = blockIdx.x;
  const int nb = bptr[b + 1] - bptr[b];
  const int nel = bs * bs;
  double acc[16];                           // bs <= 64: at most 4096 / 256 outputs per thread
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = 0.0;
  for (int pr = jd_ptr[b]; pr < jd_ptr[b + 1]; ++pr) {
    const double* jsrc = jd_val + (size_t)pr * nel;
    const double* asrc = bvinv + (size_t)jd_vblk[pr] * nel;
    for (int e = threadIdx.x; e < nel; e += 256) {
      Jd[e] = jsrc[e];
      Ai[e] = asrc[e];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nel; e += 256) {     // T = Jd * Ai
      const int i = e / bs, c = e - i * bs;
      double s = 0.0;
      for (int j = 0; j < bs; ++j) s = fma(Jd[i * bs + j], Ai[j * bs + c], s);
      T[e] = s;
    }
    __syncthreads();
    for (int e = threadIdx.x, t = 0; e < nel; e += 256, ++t) {   // acc += T * Jd^T
      const int i = e / bs, k = e - i * bs;
      double s = 0.0;
      for (int c = 0; c < bs; ++c) s = fma(T[i * bs + c], Jd[k * bs + c], s);
      acc[t] += s;
    }
    __syncthreads();
  }
  double* Bb = blocks + (size_t)b * nel;
  for (int e = threadIdx.x, t = 0; e < nel; e += 256, ++t) {
    const int i = e / bs, k = e - i * bs;
    Bb[e] = (i < nb && k < nb) ? acc[t] : (i == k ? 1.0 : 0.0);
  }
}
void launch_schur_blocks_bj(hipStream_t st, int nshift, int nblocks, int bs, const int* bptr,
                            const int* jd_ptr, const int* jd_vblk, const double* jd_val,
                            const GroupPtrs& bvinv, const GroupPtrs& blocks) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(schur_blocks_bj_kernel, dim3(nblocks, nshift), dim3(256),
                     (size_t)3 * bs * bs * sizeof(double), st, bs, bptr, jd_ptr, jd_vblk, jd_val, bvinv,
                     blocks);
}

// In-place inverse of dense bs x bs blocks by Gauss-Jordan with partial
// pivoting in LDS; one workgroup per block.  flag[0] is set to 1 on a zero pivot.
__global__ __launch_bounds__(256) void block_invert_kernel(int bs, const int* __restrict__ bptr,
                                                           GroupPtrs blockss,
                                                           int* __restrict__ flag) {
  double* __restrict__ blocks = const_cast<double*>(blockss.p[blockIdx.y]);   // blockIdx.y = shift
  extern __shared__ double sm[];  // bs x (2*bs) augmented matrix
  __shared__ int piv;
  __shared__ double pivval;
  const int W = 2 * bs;
  double* Bb = blocks + (size_t)blockIdx.x * bs * bs;
  // rows / columns beyond the block's true size are identity padding
  const int nb = bptr ? bptr[blockIdx.x + 1] - bptr[blockIdx.x] : bs;
  for (int e = threadIdx.x; e < bs * W; e += blockDim.x) {
    const int i = e / W, j = e - i * W;
    double v;
    if (j < bs)
      v = (i < nb && j < nb) ? Bb[i * bs + j] : (i == j ? 1.0 : 0.0);
    else
      v = ((j - bs) == i) ? 1.0 : 0.0;
    sm[e] = v;
  }
  __syncthreads();
  for (int k = 0; k < bs; ++k) {
    if (threadIdx.x == 0) {
      int p = k;
      double best = fabs(sm[k * W + k]);
      for (int i = k + 1; i < bs; ++i) {
        const double v = fabs(sm[i * W + k]);
        if (v > best) { best = v; p = i; }
      }
      piv = p;
      pivval = sm[p * W + k];
      if (!(best > 0.0)) { flag[0] = 1; pivval = 1.0; }
    }
    __syncthreads();
    const int p = piv;
    if (p != k) {
      for (int j = threadIdx.x; j < W; j += blockDim.x) {
        const double t = sm[k * W + j];
        sm[k * W + j] = sm[p * W + j];
        sm[p * W + j] = t;
      }
    }
    __syncthreads();
    const double ipv = 1.0 / pivval;
    for (int j = threadIdx.x; j < W; j += blockDim.x) sm[k * W + j] *= ipv;
    __syncthreads();
    for (int e = threadIdx.x; e < bs * W; e += blockDim.x) {
      const int i = e / W, j = e - i * W;
      if (i != k && j != k) sm[e] -= sm[i * W + k] * sm[k * W + j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bs; i += blockDim.x)
      if (i != k) sm[i * W + k] = 0.0;
    __syncthreads();
  }
  for (int e = threadIdx.x; e < bs * bs; e += blockDim.x) {
    const int i = e / bs, j = e - i * bs;
    Bb[e] = sm[i * W + bs + j];
  }
}
void launch_block_invert(hipStream_t st, int nshift, int nblocks, int bs, const int* bptr,
                         const GroupPtrs& blocks, int* flag) {
  if (nblocks <= 0 || nshift <= 0) return;
  hipLaunchKernelGGL(block_invert_kernel, dim3(nblocks, nshift), dim3(256),
                     (size_t)bs * 2 * bs * sizeof(double), st, bs, bptr, blocks, flag);
}

// ---------------------------------------------------------------------------
// coarse level: restriction (aggregate sums), dense apply, prolongation-add
// ---------------------------------------------------------------------------

// ec = Einv (k x k, row-major) * rc (k x m) on the FP64 matrix cores.
// A workgroup of 8 waves owns 16 output rows x 16 columns; wave w sweeps the
// k-range [w*kslice, (w+1)*kslice) in chunks of 16 columns of Einv.  Per chunk a
// lane (r = l&15, q = l>>4) loads Einv[i0+r][j0+4q .. j0+4q+3] as one 32-B
// vector (the 4 q-lanes of a row cover one 128-B line) and feeds element s to
// MFMA s; the k index of that MFMA's slot q is column j0+4q+s, so the B operand
// is rc[j0+4q+s][c].  Partial tiles are summed through LDS.
typedef double d4v __attribute__((ext_vector_type(4)));
// `ld`: leading dimension of the stored inverse (k for FP64, k rounded up to 4 for FP32)
template <class T>
__global__ __launch_bounds__(512) void dense_apply_kernel(GroupTab gt, int k, int m,
                                                          GroupPtrsT<T> Einvs, int ld,
                                                          const double* __restrict__ rc,
                                                          double* __restrict__ ec) {
  __shared__ double red[8][16][17];
  const int grp = gt.gid[blockIdx.z];
  const T* __restrict__ Einv = Einvs.p[grp];
  rc += (size_t)grp * k * m;
  ec += (size_t)grp * k * m;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int i0 = blockIdx.x * 16, c0 = blockIdx.y * 16;
  const int nchunk = (k + 15) / 16;
  const int per = (nchunk + 7) / 8;
  const int ch0 = w * per, ch1 = min(nchunk, ch0 + per);
  const int row = i0 + r;
  const int col = c0 + r;           // B / D column owned by this lane
  d4v acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
  // fetch of one 16-column chunk: 4 values of Einv (one 32-B read when aligned)
  // and the 4 matching rows of rc
  auto fetch = [&](int ch, double (&a)[4], double (&bb)[4]) {
    const int j = ch * 16 + 4 * q;
    // vector load when the 4 entries exist and are aligned to the vector size
    if (row < k && j + 3 < k && ((size_t)row * ld + j) % (sizeof(T) == 4 ? 4 : 2) == 0) {
      load4(Einv + (size_t)row * ld + j, a);
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        a[t] = (row < k && j + t < k) ? (double)Einv[(size_t)row * ld + j + t] : 0.0;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int jj = j + t;
      bb[t] = (jj < k && col < m) ? rc[(size_t)jj * m + col] : 0.0;
    }
  };
  // four chunks are requested before the first MFMA consumes any of them, so the
  // wave keeps ~20 loads in flight instead of waiting per chunk
  for (int ch = ch0; ch < ch1; ch += 4) {
    double a0[4], b0[4], a1[4], b1[4], a2[4], b2[4], a3[4], b3[4];
    fetch(ch, a0, b0);
    fetch(ch + 1 < ch1 ? ch + 1 : nchunk, a1, b1);
    fetch(ch + 2 < ch1 ? ch + 2 : nchunk, a2, b2);
    fetch(ch + 3 < ch1 ? ch + 3 : nchunk, a3, b3);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[t], b0[t], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[t], b1[t], acc2, 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[t], b2[t], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a3[t], b3[t], acc2, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] += acc2[e];
  // D[row = q + 4*e][col = r]
#pragma unroll
  for (int e = 0; e < 4; ++e) red[w][q + 4 * e][r] = acc[e];
  __syncthreads();
  if (threadIdx.x < 256) {
    const int rr = threadIdx.x >> 4, cc = threadIdx.x & 15;
    double sum = 0.0;
#pragma unroll
    for (int t = 0; t < 8; ++t) sum += red[t][rr][cc];
    if (i0 + rr < k && c0 + cc < m) ec[(size_t)(i0 + rr) * m + c0 + cc] = sum;
  }
}
void launch_dense_apply_b(hipStream_t st, const GroupTab& gt, int k, int m, const GroupPtrs& Einv,
                          const double* rc, double* ec) {
  if (k <= 0 || gt.ng <= 0) return;
  dim3 grid((k + 15) / 16, (m + 15) / 16, gt.ng);
  hipLaunchKernelGGL(dense_apply_kernel<double>, grid, dim3(512), 0, st, gt, k, m, Einv, k, rc, ec);
}
// Low-precision-stored inverse in TILE-MAJOR layout: 16 x 16 tiles of 256 contiguous entries,
// tile (it, jt) at (it * kp + jt) * 256, kp = ceil(k / 16), zero padded.  The A operand
// of one MFMA chunk -- lane (r, q) needs Einv[16 it + r][16 jt + 4q .. 4q+3] -- is then
// ONE fully coalesced 1-KB read per wave (lane offset (16 r + 4 q) entries) instead of sixteen
// row segments 4k bytes apart.  (An FP16-stored inverse with row / column scales was tried in
// round 2: +2.4 % at cfg2 with row scales, but no usable preconditioner for the mass-dominated
// DRE operator of cfg4 either way, and the column scales cost more loads than the bytes save.)
__device__ __forceinline__ void load4t(const float* p, double (&a)[4]) {
  const float4 u = *reinterpret_cast<const float4*>(p);
  a[0] = (double)u.x; a[1] = (double)u.y; a[2] = (double)u.z; a[3] = (double)u.w;
}
// Register blocking: a workgroup owns TI row tiles (16 TI output rows) x 16 columns; per
// 16-column chunk of the inverse a lane loads its 4 values of rc ONCE and feeds them to the
// MFMAs of all TI row tiles -- the rc gathers, four per chunk, were the larger part of the
// kernel's load instructions (the inverse itself is one vector load per chunk and tile).
template <class T, int TI>
__global__ __launch_bounds__(512) void dense_apply_tiled_kernel(GroupTab gt, int k, int m,
                                                                GroupPtrsT<T> Einvs,
                                                                const double* __restrict__ rc,
                                                                double* __restrict__ ec) {
  __shared__ double red[8][16 * TI][17];
  const int grp = gt.gid[blockIdx.z];
  const T* __restrict__ Einv = Einvs.p[grp];
  rc += (size_t)grp * k * m;
  ec += (size_t)grp * k * m;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int it0 = blockIdx.x * TI, c0 = blockIdx.y * 16;
  const int kp = (k + 15) / 16;
  const int per = (kp + 7) / 8;
  const int ch0 = w * per, ch1 = min(kp, ch0 + per);
  const int col = c0 + r;
  d4v acc[TI];
#pragma unroll
  for (int t = 0; t < TI; ++t) acc[t] = (d4v){0.0, 0.0, 0.0, 0.0};
  const size_t lane_off = (size_t)r * 16 + 4 * q;
  for (int ch = ch0; ch < ch1; ch += 2) {
    double b0[4], b1[4];
    const bool two = ch + 1 < ch1;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j0 = ch * 16 + 4 * q + t, j1 = j0 + 16;
      b0[t] = (j0 < k && col < m) ? rc[(size_t)j0 * m + col] : 0.0;
      b1[t] = (two && j1 < k && col < m) ? rc[(size_t)j1 * m + col] : 0.0;
    }
    double a0[TI][4], a1[TI][4];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
      const int it = it0 + ti;
      if (it < kp) {
        const T* __restrict__ trow = Einv + ((size_t)it * kp + ch) * 256 + lane_off;
        load4t(trow, a0[ti]);
        if (two) {
          load4t(trow + 256, a1[ti]);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) a1[ti][t] = 0.0;
        }
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) a0[ti][t] = a1[ti][t] = 0.0;
      }
    }
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[ti][t], b0[t], acc[ti], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[ti][t], b1[t], acc[ti], 0, 0, 0);
    }
  }
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[w][16 * ti + q + 4 * e][r] = acc[ti][e];
  __syncthreads();
  for (int o = threadIdx.x; o < 16 * TI * 16; o += 512) {
    const int rr = o >> 4, cc = o & 15;
    double sum = 0.0;
#pragma unroll
    for (int t = 0; t < 8; ++t) sum += red[t][rr][cc];
    const int row = it0 * 16 + rr;
    if (row < k && c0 + cc < m) ec[(size_t)row * m + c0 + cc] = sum;
  }
}
template <class T>
static void dense_apply_tiled_launch(hipStream_t st, const GroupTab& gt, int k, int m,
                                     const GroupPtrsT<T>& Einv, const double* rc, double* ec) {
  if (k <= 0 || gt.ng <= 0) return;
  const int kp = (k + 15) / 16;
  // one row tile per workgroup: four tiles per workgroup (rc values loaded once for four MFMA
  // groups) measured no faster -- 133 VGPRs, 3 waves per SIMD: 63 vs 57-63 us at cfg2, G = 16; two tiles
  // (81 VGPRs): 54.9 vs 52.8 us at cfg2, 168 vs 174 us on the 3.2k child matrix of cfg5; four chunks per pass
  // with all 20 loads issued together (70 VGPRs): 55.4 vs 50-52 us at cfg2, 178 vs 165 us at cfg5 -- the launch
  // wants waves, not loads per wave
  dim3 grid(kp, (m + 15) / 16, gt.ng);
  hipLaunchKernelGGL((dense_apply_tiled_kernel<T, 1>), grid, dim3(512), 0, st, gt, k, m, Einv, rc, ec);
}
void launch_dense_apply_b(hipStream_t st, const GroupTab& gt, int k, int m, const GroupPtrsF& Einv,
                          int ldf, const double* rc, double* ec) {
  (void)ldf;   // tile-major storage (launch_to_f32_tiled)
  dense_apply_tiled_launch(st, gt, k, m, Einv, rc, ec);
}
// dst = FP32 copy of the k x k row-major src in 16 x 16 tile-major layout, zero padded
__global__ void to_f32_tiled_kernel(int k, const double* __restrict__ src, float* __restrict__ dst) {
  const int kp = (k + 15) / 16;
  const size_t n = (size_t)kp * kp * 256;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t tile = i >> 8;
    const int r = (int)((i >> 4) & 15), cc = (int)(i & 15);
    const int row = (int)(tile / kp) * 16 + r, col = (int)(tile % kp) * 16 + cc;
    dst[i] = (row < k && col < k) ? (float)src[(size_t)row * k + col] : 0.f;
  }
}
void launch_to_f32_tiled(hipStream_t st, int k, const double* src, float* dst) {
  const int kp = (k + 15) / 16;
  const size_t n = (size_t)kp * kp * 256;
  if (!n) return;
  int grid = (int)std::min<size_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(to_f32_tiled_kernel, dim3(grid), dim3(256), 0, st, k, src, dst);
}


// ---------------------------------------------------------------------------
// Batched in-place inverse of the dense coarse matrices by BLOCK Gauss-Jordan elimination without
// pivoting (row-major k x k, one pointer per matrix).  Per diagonal block I (GJ_NB rows):
//   gj_prep:    Cb = A[:, I] with the rows I zeroed;  Rp = A[I, :] with the block A[I, I] replaced by the
//               identity;  D = A[I, I];  A[:, I] = 0
//   gj_diag:    D <- D^-1 (one workgroup per matrix, Gauss-Jordan in LDS)
//   two batched rocBLAS GEMMs (ricadi_solver.hip):  Rb = D^-1 Rp,   A -= Cb Rb
//   gj_rows:    A[I, :] = Rb
// after the last block A holds its inverse.  All of the 2 k^3 flops are in the rank-GJ_NB updates on the
// matrix cores (rocSOLVER's getrf + getri spend a third of their time in one poorly parallel kernel).
// ---------------------------------------------------------------------------
constexpr int GJ_NB = 128;
// up to GJ_MAX matrices per call: the 16 shifts of a sweep and the projection operator go through ONE batch
// (a matrix inverted alone costs several times its share of a batch)
constexpr int GJ_MAX = 24;
struct GjPtrs {
  double* a[GJ_MAX];
};
__global__ __launch_bounds__(256) void gj_prep_kernel(GjPtrs A, int k, int k0, int nbe, double* __restrict__ Cb,
                                                      double* __restrict__ Rp, double* __restrict__ D) {
  double* __restrict__ a = A.a[blockIdx.z];
  (void)Cb;
  double* rp = Rp + (size_t)blockIdx.z * k * GJ_NB;
  double* d = D + (size_t)blockIdx.z * GJ_NB * GJ_NB;
  const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  // row panel (nbe x k) first: it reads the diagonal block before the column pass zeroes it
  for (size_t e = tid; e < (size_t)nbe * k; e += nth) {
    const int i = (int)(e / k), j = (int)(e - (size_t)i * k);
    const double v = a[(size_t)(k0 + i) * k + j];
    const bool inb = j >= k0 && j < k0 + nbe;
    rp[(size_t)i * k + j] = inb ? (j - k0 == i ? 1.0 : 0.0) : v;
    if (inb) d[(size_t)i * GJ_NB + (j - k0)] = v;
  }
}
__global__ __launch_bounds__(256) void gj_cols_kernel(GjPtrs A, int k, int k0, int nbe, double* __restrict__ Cb) {
  double* __restrict__ a = A.a[blockIdx.z];
  double* cb = Cb + (size_t)blockIdx.z * k * GJ_NB;
  const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  for (size_t e = tid; e < (size_t)k * nbe; e += nth) {
    const int i = (int)(e / nbe), j = (int)(e - (size_t)i * nbe);
    const size_t at = (size_t)i * k + k0 + j;
    const bool inrow = i >= k0 && i < k0 + nbe;
    cb[(size_t)i * GJ_NB + j] = inrow ? 0.0 : a[at];
    a[at] = 0.0;
  }
}
// D (GJ_NB x GJ_NB, row-major, leading nbe x nbe block used) <- its inverse; flag set on a vanishing pivot.
// One workgroup of 1024 threads per matrix; the matrix lives in REGISTERS (thread (bi, bj) owns the 4 x 4
// sub-block at rows 4 bi, columns 4 bj), a Gauss-Jordan step only passes the pivot row and column through LDS
// (double buffered: one barrier per step).  The first version kept the matrix in LDS and rewrote all of it
// per step: bound by the LDS store rate, 260 us per call instead of ~25.
__global__ __launch_bounds__(1024) void gj_diag_kernel(double* __restrict__ D, int nbe, int* __restrict__ flag) {
  __shared__ double rowb[2][GJ_NB], colb[2][GJ_NB];
  double* d = D + (size_t)blockIdx.x * GJ_NB * GJ_NB;
  const int bi = threadIdx.x & 31, bj = threadIdx.x >> 5;
  const int i0 = 4 * bi, j0 = 4 * bj;
  double a[4][4];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
      const int i = i0 + ti, j = j0 + tj;
      a[ti][tj] = (i < nbe && j < nbe) ? d[(size_t)i * GJ_NB + j] : (i == j ? 1.0 : 0.0);
    }
  // scale of the block: pivots are judged RELATIVE to the largest entry (a tiny but non-zero pivot would
  // otherwise pass and leave a garbage inverse behind -- GMRES then stalls instead of the pivoted route running)
  __shared__ double wmax[16];
  double amax = 0.0;
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
      if (i0 + ti < nbe && j0 + tj < nbe) amax = fmax(amax, fabs(a[ti][tj]));
  for (int off = 32; off > 0; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = 0.0;
#pragma unroll
  for (int w = 0; w < 16; ++w) amax = fmax(amax, wmax[w]);
  const double ptol = 1e-12 * amax;
  bool bad = !(amax > 0.0) || !(amax < 1e300);
  for (int p = 0; p < nbe && !bad; ++p) {
    const int buf = p & 1;
    if ((p >> 2) == bi) {          // owners of the pivot row
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
        if (i0 + ti == p) {
#pragma unroll
          for (int tj = 0; tj < 4; ++tj) rowb[buf][j0 + tj] = a[ti][tj];
        }
    }
    if ((p >> 2) == bj) {          // owners of the pivot column
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
        if (j0 + tj == p) {
#pragma unroll
          for (int ti = 0; ti < 4; ++ti) colb[buf][i0 + ti] = a[ti][tj];
        }
    }
    __syncthreads();
    const double piv = rowb[buf][p];
    if (!(fabs(piv) > ptol)) {     // uniform: every thread reads the same pivot
      bad = true;
      break;
    }
    const double inv = 1.0 / piv;
    double rj[4], ci[4];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) rj[tj] = (j0 + tj == p) ? inv : rowb[buf][j0 + tj] * inv;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) ci[ti] = colb[buf][i0 + ti];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        if (i0 + ti == p) a[ti][tj] = rj[tj];
        else a[ti][tj] = ((j0 + tj == p) ? 0.0 : a[ti][tj]) - ci[ti] * rj[tj];
      }
  }
  if (bad && threadIdx.x == 0) atomicExch(flag, 1);
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) d[(size_t)(i0 + ti) * GJ_NB + j0 + tj] = a[ti][tj];
}
__global__ __launch_bounds__(256) void gj_rows_kernel(GjPtrs A, int k, int k0, int nbe, const double* __restrict__ Rb) {
  double* __restrict__ a = A.a[blockIdx.z];
  const double* rb = Rb + (size_t)blockIdx.z * k * GJ_NB;
  const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  for (size_t e = tid; e < (size_t)nbe * k; e += nth) a[(size_t)k0 * k + e] = rb[e];
}
int gj_block() { return GJ_NB; }
int gj_max_batch() { return GJ_MAX; }
void launch_gj_prep(hipStream_t st, int nb, double* const* mats, int k, int k0, int nbe, double* Cb, double* Rp,
                    double* D) {
  GjPtrs P;
  for (int i = 0; i < GJ_MAX; ++i) P.a[i] = i < nb ? mats[i] : nullptr;
  const int grid = (int)std::min<size_t>(((size_t)nbe * k + 255) / 256, 1024);
  hipLaunchKernelGGL(gj_prep_kernel, dim3(grid, 1, nb), dim3(256), 0, st, P, k, k0, nbe, Cb, Rp, D);
  hipLaunchKernelGGL(gj_cols_kernel, dim3(grid, 1, nb), dim3(256), 0, st, P, k, k0, nbe, Cb);
}
void launch_gj_diag(hipStream_t st, int nb, double* D, int nbe, int* flag) {
  hipLaunchKernelGGL(gj_diag_kernel, dim3(nb), dim3(1024), 0, st, D, nbe, flag);
}
void launch_gj_rows(hipStream_t st, int nb, double* const* mats, int k, int k0, int nbe, const double* Rb) {
  GjPtrs P;
  for (int i = 0; i < GJ_MAX; ++i) P.a[i] = i < nb ? mats[i] : nullptr;
  const int grid = (int)std::min<size_t>(((size_t)nbe * k + 255) / 256, 1024);
  hipLaunchKernelGGL(gj_rows_kernel, dim3(grid, 1, nb), dim3(256), 0, st, P, k, k0, nbe, Rb);
}

// dst (FP32, leading dimension ldd) = src (FP64, leading dimension lds_)
__global__ void to_f32_kernel(int nrows, int ncols, const double* __restrict__ src, int lds_,
                              float* __restrict__ dst, int ldd) {
  const size_t n = (size_t)nrows * ldd;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / ldd;
    const int c = (int)(i - r * ldd);
    dst[i] = c < ncols ? (float)src[r * lds_ + c] : 0.f;
  }
}
void launch_to_f32(hipStream_t st, int nrows, int ncols, const double* src, int lds_, float* dst,
                   int ldd) {
  const size_t n = (size_t)nrows * ldd;
  if (!n) return;
  int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(to_f32_kernel, dim3(grid), dim3(256), 0, st, nrows, ncols, src, lds_, dst, ldd);
}


// ---------------------------------------------------------------------------
// K5/K6: dense tall-skinny products on the FP64 matrix cores.
//   v_mfma_f64_16x16x4_f64: lane l holds A[i = l&15][k = l>>4] and
//   B[k = l>>4][j = l&15]; D[row = (l>>4) + 4*reg][col = l&15].
//
// gemm_tn:  C (p x q) += A^T B, A n x p, B n x q (row-major).  Both operands
// are read as 4-row x 16-column slabs -> each 16-lane group reads one 128-B
// line.  A wave owns TI x TJ tiles of C over a row range; partial results are
// added with FP64 atomics (C must be zeroed by the caller).
// ---------------------------------------------------------------------------
template <int TI, int TJ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_tn_kernel(GroupTab gt, int ptiles, int n, int p, int q,
                                                      const double* __restrict__ A, int lda,
                                                      const double* __restrict__ B, int ldb,
                                                      size_t gsB, double* __restrict__ C, int ldc,
                                                      size_t gsC, int rows_per_wave, int symmetric, int combine) {
  // batched form: blockIdx.y = (active group) * ptiles + (tile row); A is shared
  const int by = blockIdx.y % ptiles;
  {
    const int grp = gt.gid[blockIdx.y / ptiles];
    B += (size_t)grp * gsB;
    C += (size_t)grp * gsC;
  }
  // symmetric (A == B, Gram matrix): only tile blocks on / above the diagonal
  // are computed, the strictly upper ones are mirrored when written
  if (symmetric && blockIdx.z < by) return;
  const int lane = threadIdx.x & 63;
  const int wave_in_blk = threadIdx.x >> 6;
  const int i0 = by * 16 * TI;
  const int j0 = blockIdx.z * 16 * TJ;
  const int rbeg = (blockIdx.x * 4 + wave_in_blk) * rows_per_wave;
  const int rend = min(n, rbeg + rows_per_wave);   // empty range for surplus waves (they still join the barrier)
  const int lc = lane & 15, lk = lane >> 4;
  d4 acc[TI][TJ];
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  // operands of the NEXT 4-row step are loaded before the MFMAs of the current one (a wide product runs about one
  // wave per SIMD: without the prefetch every step waited a full memory round trip for its 8 loads)
  double af[TI], bf[TJ], afn[TI], bfn[TJ];
  auto load4 = [&](int r, double (&fa)[TI], double (&fb)[TJ]) {
    const int rr = r + lk;
    const bool rok = rr < rend;
#pragma unroll
    for (int a = 0; a < TI; ++a) {
      const int col = i0 + 16 * a + lc;
      fa[a] = (rok && col < p) ? A[(size_t)rr * lda + col] : 0.0;
    }
#pragma unroll
    for (int b = 0; b < TJ; ++b) {
      const int col = j0 + 16 * b + lc;
      fb[b] = (rok && col < q) ? B[(size_t)rr * ldb + col] : 0.0;
    }
  };
  if (rbeg < rend) load4(rbeg, af, bf);
  for (int r = rbeg; r < rend; r += 4) {
    load4(r + 4, afn, bfn);                  // rows >= rend read as zeros
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
      for (int b = 0; b < TJ; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < TI; ++a) af[a] = afn[a];
#pragma unroll
    for (int b = 0; b < TJ; ++b) bf[b] = bfn[b];
  }
  if (TI * TJ > 4) {
    // wide products: the atomics are spread over many outputs; every wave adds its own tiles -- unless the
    // product has FEW tiles and is therefore cut into many short row slices (combine != 0): then the four waves
    // of the workgroup (four consecutive slices of the same tiles) are summed first, one after the other through
    // one 32-KB LDS buffer, and wave 0 alone issues the atomics (4x fewer)
    if (combine) {
      __shared__ double comb[(TI * TJ > 4) ? TI * TJ * 4 * 64 : 1];
      for (int wsrc = 1; wsrc < 4; ++wsrc) {
        if (wave_in_blk == wsrc) {
#pragma unroll
          for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int b = 0; b < TJ; ++b)
#pragma unroll
              for (int e = 0; e < 4; ++e) comb[((a * TJ + b) * 4 + e) * 64 + lane] = acc[a][b][e];
        }
        __syncthreads();
        if (wave_in_blk == 0) {
#pragma unroll
          for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int b = 0; b < TJ; ++b)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[a][b][e] += comb[((a * TJ + b) * 4 + e) * 64 + lane];
        }
        __syncthreads();
      }
      if (wave_in_blk != 0) return;
    }
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
      for (int b = 0; b < TJ; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = i0 + 16 * a + lk + 4 * e;
          const int col = j0 + 16 * b + lc;
          if ((rbeg < n || combine) && row < p && col < q) {
            atomicAdd(&C[(size_t)row * ldc + col], acc[a][b][e]);
            if (symmetric && blockIdx.z > by)
              atomicAdd(&C[(size_t)col * ldc + row], acc[a][b][e]);
          }
        }
    return;
  }
  // thin products: combine the block's four partial tile sets in LDS, then one
  // atomic per output element and block (4x fewer contended atomics)
  constexpr int NT = (TI * TJ > 4) ? 1 : TI * TJ * 4;
  __shared__ double red[4][NT][64];
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave_in_blk][((a * TJ + b) * 4 + e) % NT][lane] = acc[a][b][e];
  __syncthreads();
  if (wave_in_blk == 0) {
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
      for (int b = 0; b < TJ; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int idx = ((a * TJ + b) * 4 + e) % NT;
          const double v = (red[0][idx][lane] + red[1][idx][lane]) + (red[2][idx][lane] + red[3][idx][lane]);
          const int row = i0 + 16 * a + lk + 4 * e;
          const int col = j0 + 16 * b + lc;
          if (row < p && col < q) {
            atomicAdd(&C[(size_t)row * ldc + col], v);
            if (symmetric && blockIdx.z > by) atomicAdd(&C[(size_t)col * ldc + row], v);
          }
        }
  }
}
void launch_gemm_tn(hipStream_t st, int n, int p, int q, const double* A, int lda, const double* B,
                    int ldb, double* C, int ldc) {
  launch_gemm_tn_b(st, single_group(), n, p, q, A, lda, B, ldb, 0, C, ldc, 0);
}
void launch_gemm_tn_b(hipStream_t st, const GroupTab& gt, int n, int p, int q, const double* A,
                      int lda, const double* B, int ldb, size_t gsB, double* C, int ldc, size_t gsC) {
  if (n <= 0 || p <= 0 || q <= 0 || gt.ng <= 0) return;
  const int symmetric = (gt.ng == 1 && A == B && lda == ldb && p == q) ? 1 : 0;
  // wide products: 64 x 64 output per wave (16 MFMAs per 8 loaded operands);
  // thin ones (low-rank term, gain): 32 x 32
  const bool wide = p >= 128 && q >= 128;
  const int tp = wide ? 64 : 32;
  const int tp_ = (p + tp - 1) / tp, tq_ = (q + tp - 1) / tp;
  const int tiles = symmetric ? tp_ * (tp_ + 1) / 2 : tp_ * tq_;
  // Row slices: every slice adds its partial tiles with atomics, so wide products
  // (many output elements) take few, long slices -- about 1.5 waves per SIMD in
  // total -- while thin ones take many short slices to fill the chip.
  // wide products with few tiles (a 128-column QR panel against itself or against the earlier columns): short
  // slices fill the chip, the workgroup's four partial tile sets are combined in LDS before the atomics
  const int combine = (wide && tiles * gt.ng <= 16) ? 1 : 0;
  const int min_rows = wide ? (combine ? 64 : 256) : 64;
  const int target_waves = wide ? (combine ? 3072 : 1536) : 8192;
  int slices = std::max(1, std::min((n + min_rows - 1) / min_rows,
                                    std::max(1, target_waves / std::max(1, tiles * gt.ng))));
  int rows_per_wave = (n + slices - 1) / slices;
  rows_per_wave = (rows_per_wave + 3) & ~3;
  slices = (n + rows_per_wave - 1) / rows_per_wave;
  dim3 grid((slices + 3) / 4, tp_ * gt.ng, tq_), block(256);
  if (wide)
    hipLaunchKernelGGL((gemm_tn_kernel<4, 4>), grid, block, 0, st, gt, tp_, n, p, q, A, lda, B, ldb,
                       gsB, C, ldc, gsC, rows_per_wave, symmetric, combine);
  else
    hipLaunchKernelGGL((gemm_tn_kernel<2, 2>), grid, block, 0, st, gt, tp_, n, p, q, A, lda, B, ldb,
                       gsB, C, ldc, gsC, rows_per_wave, symmetric, 0);
}

// gemm_nn:  Y (n x q) = alpha * A (n x p) * C (p x q) + beta * Y.
// A wave owns 16 rows x (16*TJ) columns.  A-operand: A[r0 + (l&15)][k + (l>>4)].
template <int TJ>
__global__ __launch_bounds__(256) void gemm_nn_kernel(GroupTab gt, int n, int p, int q,
                                                      GroupPtrs As, int lda,
                                                      const double* __restrict__ C, int ldc,
                                                      size_t gsC, double* __restrict__ Y, int ldy,
                                                      size_t gsY, double alpha, double beta) {
  const double* __restrict__ A;
  {
    const int grp = gt.gid[blockIdx.z];
    A = As.p[grp];
    C += (size_t)grp * gsC;
    Y += (size_t)grp * gsY;
  }
  const int lane = threadIdx.x & 63;
  const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
  const int j0 = blockIdx.y * 16 * TJ;
  if (r0 >= n) return;
  const int lc = lane & 15, lk = lane >> 4;
  d4 acc[TJ];
#pragma unroll
  for (int b = 0; b < TJ; ++b) acc[b] = (d4){0.0, 0.0, 0.0, 0.0};
  const int arow = r0 + lc;
  const bool aok = arow < n;
  for (int k = 0; k < p; k += 4) {
    const int kk = k + lk;
    const double af = (aok && kk < p) ? A[(size_t)arow * lda + kk] : 0.0;
#pragma unroll
    for (int b = 0; b < TJ; ++b) {
      const int col = j0 + 16 * b + lc;
      const double bf = (kk < p && col < q) ? C[(size_t)kk * ldc + col] : 0.0;
      acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc[b], 0, 0, 0);
    }
  }
#pragma unroll
  for (int b = 0; b < TJ; ++b)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = r0 + lk + 4 * e;
      const int col = j0 + 16 * b + lc;
      if (row < n && col < q) {
        double* y = &Y[(size_t)row * ldy + col];
        *y = alpha * acc[b][e] + (beta == 0.0 ? 0.0 : beta * (*y));
      }
    }
}
void launch_gemm_nn_bp(hipStream_t st, const GroupTab& gt, int n, int p, int q, const GroupPtrs& A,
                       int lda, const double* C, int ldc, size_t gsC, double* Y, int ldy, size_t gsY,
                       double alpha, double beta) {
  if (n <= 0 || q <= 0 || gt.ng <= 0) return;
  dim3 grid((n + 63) / 64, (q + 31) / 32, gt.ng), block(256);
  hipLaunchKernelGGL((gemm_nn_kernel<2>), grid, block, 0, st, gt, n, p, q, A, lda, C, ldc, gsC, Y,
                     ldy, gsY, alpha, beta);
}
void launch_gemm_nn_b(hipStream_t st, const GroupTab& gt, int n, int p, int q, const double* A,
                      int lda, const double* C, int ldc, size_t gsC, double* Y, int ldy, size_t gsY,
                      double alpha, double beta) {
  launch_gemm_nn_bp(st, gt, n, p, q, same_ptr(A), lda, C, ldc, gsC, Y, ldy, gsY, alpha, beta);
}
void launch_gemm_nn(hipStream_t st, int n, int p, int q, const double* A, int lda, const double* C,
                    int ldc, double* Y, int ldy, double alpha, double beta) {
  launch_gemm_nn_b(st, single_group(), n, p, q, A, lda, C, ldc, 0, Y, ldy, 0, alpha, beta);
}

// ---------------------------------------------------------------------------
// K5: Householder TSQR of a tall n x w panel (w <= 32).
//
// tsqr_local: one workgroup per block of TSQR_RB rows.  The block is copied into
// LDS and factorised by w Householder reflectors (LAPACK conventions: v_k(k) = 1,
// H_k = I - tau_k v_k v_k^T); the upper triangle is the block's R.  The explicit
// thin Q of the block (TSQR_RB x w) is then formed in a second LDS tile by
// applying the reflectors in reverse order to [I; 0] -- Householder quality,
// also for (numerically) rank-deficient panels, unlike Q = A R^-1.
// The host stacks the R factors and repeats until one block is left; tsqr_apply
// multiplies the local Q's by the 32 x 32 blocks of the next level's Q on the
// way down.
// ---------------------------------------------------------------------------
constexpr int TSQR_RB = 256, TSQR_W = 32, TSQR_LD = TSQR_W + 1;

__device__ __forceinline__ double tsqr_block_sum(double v, double* red, int tid) {
  // sum over the 256 threads of the workgroup (4 waves)
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  const double s = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return s;
}

__global__ __launch_bounds__(256) void tsqr_local_kernel(int nrows, int w,
                                                         const double* __restrict__ A, int lda,
                                                         double* __restrict__ Qloc,
                                                         double* __restrict__ Rstack) {
  extern __shared__ double sm[];
  double* T = sm;                               // TSQR_RB x TSQR_LD  (the panel block)
  double* E = sm + TSQR_RB * TSQR_LD;           // TSQR_RB x TSQR_LD  (explicit Q)
  double* wsum = E + TSQR_RB * TSQR_LD;         // 8 x TSQR_W partial dot products
  double* tau = wsum + 8 * TSQR_W;              // TSQR_W
  double* red = tau + TSQR_W;                   // 4
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * TSQR_RB;
  const int nr = min(TSQR_RB, nrows - r0);
  for (int e = tid; e < TSQR_RB * TSQR_W; e += 256) {
    const int i = e / TSQR_W, j = e - i * TSQR_W;
    T[i * TSQR_LD + j] = (i < nr && j < w) ? A[(size_t)(r0 + i) * lda + j] : 0.0;
  }
  __syncthreads();
  const int jc = tid & 31, sl = tid >> 5;       // column / 32-row slice owned in the updates
  for (int k = 0; k < w; ++k) {
    // reflector for column k
    double part = 0.0;
    for (int i = k + 1 + tid; i < TSQR_RB; i += 256) part += T[i * TSQR_LD + k] * T[i * TSQR_LD + k];
    const double ssq = tsqr_block_sum(part, red, tid);
    const double alpha = T[k * TSQR_LD + k];
    double tk = 0.0, scale = 0.0, beta = alpha;
    if (ssq > 0.0) {
      beta = -copysign(sqrt(alpha * alpha + ssq), alpha);
      tk = (beta - alpha) / beta;
      scale = 1.0 / (alpha - beta);
    }
    __syncthreads();
    for (int i = k + 1 + tid; i < TSQR_RB; i += 256) T[i * TSQR_LD + k] *= scale;
    if (tid == 0) {
      T[k * TSQR_LD + k] = beta;
      tau[k] = tk;
    }
    __syncthreads();
    // trailing update: w_j = v^T T[:, j], T[:, j] -= tau v w_j   (j > k)
    double ps = 0.0;
    if (jc > k && jc < w) {
      for (int i = max(k, sl * 32); i < sl * 32 + 32; ++i) {
        const double vi = (i == k) ? 1.0 : T[i * TSQR_LD + k];
        ps = fma(vi, T[i * TSQR_LD + jc], ps);
      }
    }
    wsum[sl * TSQR_W + jc] = ps;
    __syncthreads();
    if (jc > k && jc < w) {
      double wj = 0.0;
#pragma unroll
      for (int t = 0; t < 8; ++t) wj += wsum[t * TSQR_W + jc];
      wj *= tk;
      for (int i = max(k, sl * 32); i < sl * 32 + 32; ++i) {
        const double vi = (i == k) ? 1.0 : T[i * TSQR_LD + k];
        T[i * TSQR_LD + jc] = fma(-vi, wj, T[i * TSQR_LD + jc]);
      }
    }
    __syncthreads();
  }
  // R of this block
  for (int e = tid; e < TSQR_W * TSQR_W; e += 256) {
    const int i = e / TSQR_W, j = e - i * TSQR_W;
    Rstack[(size_t)blockIdx.x * TSQR_W * TSQR_W + e] = (j >= i && i < w && j < w) ? T[i * TSQR_LD + j] : 0.0;
  }
  // explicit Q = H_0 ... H_{w-1} [I; 0]
  for (int e = tid; e < TSQR_RB * TSQR_W; e += 256) {
    const int i = e / TSQR_W, j = e - i * TSQR_W;
    E[i * TSQR_LD + j] = (i == j && j < w) ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int k = w - 1; k >= 0; --k) {
    double ps = 0.0;
    if (jc < w) {
      for (int i = max(k, sl * 32); i < sl * 32 + 32; ++i) {
        const double vi = (i == k) ? 1.0 : T[i * TSQR_LD + k];
        ps = fma(vi, E[i * TSQR_LD + jc], ps);
      }
    }
    wsum[sl * TSQR_W + jc] = ps;
    __syncthreads();
    if (jc < w) {
      double wj = 0.0;
#pragma unroll
      for (int t = 0; t < 8; ++t) wj += wsum[t * TSQR_W + jc];
      wj *= tau[k];
      for (int i = max(k, sl * 32); i < sl * 32 + 32; ++i) {
        const double vi = (i == k) ? 1.0 : T[i * TSQR_LD + k];
        E[i * TSQR_LD + jc] = fma(-vi, wj, E[i * TSQR_LD + jc]);
      }
    }
    __syncthreads();
  }
  for (int e = tid; e < nr * TSQR_W; e += 256) {
    const int i = e / TSQR_W, j = e - i * TSQR_W;
    Qloc[(size_t)(r0 + i) * TSQR_W + j] = E[i * TSQR_LD + j];
  }
}
int tsqr_num_blocks(int nrows) { return (nrows + TSQR_RB - 1) / TSQR_RB; }
void launch_tsqr_local(hipStream_t st, int nrows, int w, const double* A, int lda, double* Qloc,
                       double* Rstack) {
  const size_t lds = (size_t)(2 * TSQR_RB * TSQR_LD + 8 * TSQR_W + TSQR_W + 8) * sizeof(double);
  static bool attr_set = false;   // 137 KB of dynamic LDS: above the 64 KB default limit
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tsqr_local_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(tsqr_local_kernel, dim3(tsqr_num_blocks(nrows)), dim3(256), lds, st, nrows, w,
                     A, lda, Qloc, Rstack);
}

// Q[rows of block b] <- Qloc[rows of block b] (RB x 32) * G[b*32 .. b*32+31][:] (32 x 32);
// a block of the lower level consists of 8 stacked R's, i.e. row block b of the
// lower level's matrix corresponds to rows b*32.. of the upper level's Q.
__global__ __launch_bounds__(256) void tsqr_apply_kernel(int nrows, int w,
                                                         const double* __restrict__ Qloc,
                                                         const double* __restrict__ G,
                                                         double* __restrict__ Qout, int ldq) {
  __shared__ double g[TSQR_W][TSQR_W + 1];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int e = tid; e < TSQR_W * TSQR_W; e += 256)
    g[e / TSQR_W][e % TSQR_W] = G[(size_t)b * TSQR_W * TSQR_W + e];
  __syncthreads();
  const int r0 = b * TSQR_RB;
  const int nr = min(TSQR_RB, nrows - r0);
  for (int e = tid; e < nr * TSQR_W; e += 256) {
    const int i = e / TSQR_W, j = e - i * TSQR_W;
    const double* q = Qloc + (size_t)(r0 + i) * TSQR_W;
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < TSQR_W; ++t) s = fma(q[t], g[t][j], s);
    if (j < w) Qout[(size_t)(r0 + i) * ldq + j] = s;   // the destination may be only w columns wide
  }
}
void launch_tsqr_apply(hipStream_t st, int nrows, int w, const double* Qloc, const double* G,
                       double* Qout, int ldq) {
  hipLaunchKernelGGL(tsqr_apply_kernel, dim3(tsqr_num_blocks(nrows)), dim3(256), 0, st, nrows, w,
                     Qloc, G, Qout, ldq);
}

// ---------------------------------------------------------------------------
// K5, fast panel factorisation: Cholesky QR on the matrix cores.
//
// For a tall n x w panel P (w <= 32) whose column-normalised Gram matrix is safely
// positive definite, two rounds of
//     G = P^T P (MFMA),  Ghat = D G D = L L^T  (D = diag(G)^-1/2),  Q = P (D L^-T) (MFMA)
// give Householder-quality orthogonality (CholQR2) at GEMM speed: the 32 x 32 part below is
// this one-wave kernel, everything tall is gemm_tn / gemm_nn.  The kernel raises `flag` --
// the caller then falls back to the Householder TSQR tree -- when a normalised pivot drops
// below 1e-12 (cond(P D) beyond ~1e6: the second round could not repair the first) or a
// column is exactly zero; it never produces Inf / NaN (a failed pivot is replaced by 1, the
// column's transformation by 0).
//   in : G (32 x 32, ld 32; only the leading w x w block is meaningful), Rprev (or NULL)
//   out: T (32 x 32): Q = P T;   R (32 x 32 upper): P = Q R for the first round, and
//        R = R_this * Rprev for the second (Rprev = first round's R)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void cholqr_small_kernel(int w, const double* __restrict__ G,
                                                          const double* __restrict__ Rprev,
                                                          double* __restrict__ T, double* __restrict__ R,
                                                          int* __restrict__ flag) {
  __shared__ double A[32][33];      // Ghat, then L (lower)
  __shared__ double X[32][33];      // U^-1, U = L^T
  __shared__ double dsc[32], dinv[32];
  const int lane = threadIdx.x;
  bool bad = false;
  if (lane < 32) {
    const double gjj = lane < w ? G[lane * 32 + lane] : 1.0;
    if (lane < w && !(gjj > 0.0)) bad = true;
    dsc[lane] = (lane < w && gjj > 0.0) ? 1.0 / sqrt(gjj) : 0.0;
    dinv[lane] = (lane < w && gjj > 0.0) ? sqrt(gjj) : 0.0;
  }
  __syncthreads();
  for (int e = lane; e < 1024; e += 64) {
    const int i = e >> 5, j = e & 31;
    double v;
    if (i < w && j < w && dsc[i] > 0.0 && dsc[j] > 0.0)
      v = dsc[i] * dsc[j] * G[i * 32 + j];
    else
      v = (i == j) ? 1.0 : 0.0;      // identity padding (columns beyond w, zero columns)
    A[i][j] = v;
    X[i][j] = 0.0;
  }
  __syncthreads();
  // right-looking Cholesky, one wave
  for (int k = 0; k < 32; ++k) {
    double piv = A[k][k];
    if (!(piv > 1e-12)) {
      bad = true;
      piv = 1.0;
    }
    const double lkk = sqrt(piv);
    __syncthreads();
    if (lane == k) A[k][k] = lkk;
    if (lane > k && lane < 32) A[lane][k] /= lkk;
    __syncthreads();
    for (int e = lane; e < 1024; e += 64) {
      const int i = e >> 5, j = e & 31;
      if (j > k && i >= j) A[i][j] -= A[i][k] * A[j][k];
    }
    __syncthreads();
  }
  // X = U^-1 with U = L^T (upper): column j by back substitution, one lane per column
  if (lane < 32) {
    const int j = lane;
    X[j][j] = 1.0 / A[j][j];
    for (int i = j - 1; i >= 0; --i) {
      double sacc = 0.0;
      for (int t = i + 1; t <= j; ++t) sacc = fma(A[t][i], X[t][j], sacc);   // U[i][t] = L[t][i]
      X[i][j] = -sacc / A[i][i];
    }
  }
  __syncthreads();
  // T = D X (zero for failed / padded columns), Rcur = U D^-1
  for (int e = lane; e < 1024; e += 64) {
    const int i = e >> 5, j = e & 31;
    const bool okc = j < w && dsc[j] > 0.0;
    T[e] = (okc && i <= j) ? dsc[i] * X[i][j] : 0.0;
    X[i][j] = (i <= j && i < w && okc) ? A[j][i] * dinv[j] : 0.0;   // X now holds Rcur
  }
  __syncthreads();
  for (int e = lane; e < 1024; e += 64) {
    const int i = e >> 5, j = e & 31;
    double v;
    if (Rprev) {
      v = 0.0;
      for (int t = i; t <= j; ++t) v = fma(X[i][t], Rprev[t * 32 + j], v);
    } else {
      v = X[i][j];
    }
    R[e] = (i <= j) ? v : 0.0;
  }
  if (__any(bad) && lane == 0) atomicExch(flag, 1);
}
void launch_cholqr_small(hipStream_t st, int w, const double* G, const double* Rprev, double* T,
                         double* R, int* flag) {
  hipLaunchKernelGGL(cholqr_small_kernel, dim3(1), dim3(64), 0, st, w, G, Rprev, T, R, flag);
}

// ---------------------------------------------------------------------------
// K5w: the same panel step for panels of up to 128 columns, one workgroup of 256 threads, the
// 128 x 128 Gram matrix in LDS (132 KB): blocked right-looking Cholesky in 16-column blocks
//   - the 16 x 16 diagonal block is factorised AND inverted by wave 0 alone, one lane per row, rows in
//     registers, cross-lane operands by shuffles (no workgroup barrier inside);
//   - panel  L21 = A21 L11^-T  one thread per row against the explicit 16 x 16 inverse;
//   - trailing update, 7 x 7 entries per thread;
// then R = L^T D^-1 goes out and L is overwritten, block row by block row, by its
// inverse  X[I,J] = -X[I,I] sum_K L[I,K] X[K,J]  (the inverses of the diagonal blocks wait transposed in the
// unused upper triangle of those blocks until L's diagonal blocks are no longer needed); T = D X^T.
// Same conventions and breakdown flag as cholqr_small_kernel, except that the second round's
// R_this * R_prev is left to the caller; G has leading dimension ldg, T and R are 128 x 128 (ld 128).
// ---------------------------------------------------------------------------
constexpr int CQW = 128, CQLD = 129;
#ifdef RICADI_CQ_TIMING
#define CQT(i) long long cqt##i = wall_clock64()
#define CQA(i) do { long long t_ = wall_clock64(); if (i > 0) cqa[i - 1] += t_ - cql; cql = t_; } while (0)
#define CQP() if (tid == 0) printf("cholqr_wide w=%d: load %lld chol %lld [diag %lld panel %lld trail %lld] Rout %lld move %lld inv %lld T %lld (x10ns)\n", w, cqt1-cqt0, cqt2-cqt1, cqa[0], cqa[1], cqa[2], cqt3-cqt2, cqt4-cqt3, cqt5-cqt4, cqt6-cqt5)
#else
#define CQT(i)
#define CQA(i)
#define CQP()
#endif
__global__ __launch_bounds__(256) void cholqr_wide_kernel(int w, const double* __restrict__ G, int ldg,
                                                          double* __restrict__ T, double* __restrict__ R,
                                                          int* __restrict__ flag) {
  extern __shared__ double sm[];
  double* A = sm;                          // CQW x CQLD
  double* dsc = A + CQW * CQLD;            // 128: D
  double* dinv = dsc + CQW;                // 128: D^-1
  double* xd = dinv + CQW;                 // 128: diagonal of L^-1
  double* tmp = xd + CQW;                  // 16 x 112: S of a block row
  const int tid = threadIdx.x, lane = tid & 63;
  const int nbk = (w + 15) >> 4, wp = nbk * 16;      // blocks / padded width actually worked on
  bool bad = false;
#ifdef RICADI_CQ_TIMING
  long long cqa[3] = {0, 0, 0}, cql = 0;
#endif
  CQT(0);
  if (tid < CQW) {
    const double gjj = tid < w ? G[(size_t)tid * ldg + tid] : 1.0;
    if (tid < w && !(gjj > 0.0)) bad = true;
    dsc[tid] = (tid < w && gjj > 0.0) ? 1.0 / sqrt(gjj) : 0.0;
    dinv[tid] = (tid < w && gjj > 0.0) ? sqrt(gjj) : 0.0;
  }
  __syncthreads();
  {
    // thread = (row strip tid >> 5, column tid & 31 (+32 q)): raw loads first (16 rows in flight), then the scaling
    const int jj = tid & 31, is = tid >> 5;
    for (int j = jj; j < wp; j += 32)
      for (int ib = is; ib < wp; ib += 128) {
        double g[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int i = ib + 8 * q;
          g[q] = (i < w && j < w) ? G[(size_t)i * ldg + j] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int i = ib + 8 * q;
          if (i < wp) {
            double v;
            if (i < w && j < w && dsc[i] > 0.0 && dsc[j] > 0.0)
              v = dsc[i] * dsc[j] * g[q];
            else
              v = (i == j) ? 1.0 : 0.0;        // identity padding (columns beyond w, zero columns)
            A[i * CQLD + j] = v;
          }
        }
      }
  }
  __syncthreads();
  CQT(1);
  for (int kb = 0; kb < nbk; ++kb) {
    const int k0 = kb * 16;
    CQA(0);
    if (tid < 64) {
      // wave 0: lane i (< 16) owns row i of the diagonal block; only the factor here -- the block's inverse is
      // formed after the factorisation, all blocks at once (cholqr_diag_inverse)
      const int i = lane & 15;
      double r[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) r[j] = A[(k0 + i) * CQLD + k0 + j];
      double myinv = 0.0;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        double piv = __shfl(r[t], t, 64);
        if (!(piv > 1e-12)) {
          bad = true;
          piv = 1.0;
        }
        // 1 / l_tt by the hardware estimate + two Newton steps (no sqrt / division in the dependent chain;
        // the second CholQR round absorbs the last-bit differences)
        double rs = __builtin_amdgcn_rsq(piv);
        rs = rs * fma(-0.5 * piv * rs, rs, 1.5);
        rs = rs * fma(-0.5 * piv * rs, rs, 1.5);
        if (i == t) myinv = rs;
        r[t] = (i == t) ? piv * rs : r[t] * rs;        // rows i < t hold upper entries nobody reads
#pragma unroll
        for (int j = t + 1; j < 16; ++j) r[j] = fma(-r[t], __shfl(r[t], j, 64), r[j]);
      }
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (j <= i) A[(k0 + i) * CQLD + k0 + j] = r[j];               // L11 (lower incl. diagonal)
        xd[k0 + i] = myinv;
      }
    }
    __syncthreads();
    CQA(1);
    // panel: L21 = A21 L11^-T by forward substitution, one thread per row below the block
    {
      const int i = k0 + 16 + tid;
      if (i < wp) {
        double l[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          double sacc = A[i * CQLD + k0 + j];
#pragma unroll
          for (int k = 0; k < j; ++k) sacc = fma(-l[k], A[(k0 + j) * CQLD + k0 + k], sacc);   // L11[j][k]
          l[j] = sacc * xd[k0 + j];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) A[i * CQLD + k0 + j] = l[j];
      }
    }
    __syncthreads();
    CQA(2);
    // trailing update of the lower triangle behind the block: thread (ti, tj) owns the entries
    // (r0 + ti + 16 a, r0 + tj + 16 b), b <= a, in registers; 14 LDS reads per 28 (49) products
    {
      const int ti = tid >> 4, tj = tid & 15, r0 = k0 + 16;
      const int nt = (wp - r0) >> 4;                 // 16-row strips behind the block (wp is a multiple of 16)
      if (nt > 0) {
        double acc[7][7];
#pragma unroll
        for (int a = 0; a < 7; ++a)
#pragma unroll
          for (int b = 0; b < 7; ++b) acc[a][b] = 0.0;
#pragma unroll 4
        for (int t = 0; t < 16; ++t) {
          double li[7], lj[7];
#pragma unroll
          for (int a = 0; a < 7; ++a) {
            li[a] = a < nt ? A[(r0 + ti + 16 * a) * CQLD + k0 + t] : 0.0;
            lj[a] = a < nt ? A[(r0 + tj + 16 * a) * CQLD + k0 + t] : 0.0;
          }
#pragma unroll
          for (int a = 0; a < 7; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) acc[a][b] = fma(li[a], lj[b], acc[a][b]);
        }
#pragma unroll
        for (int a = 0; a < 7; ++a)
#pragma unroll
          for (int b = 0; b <= a; ++b)
            if (a < nt && (b < a || tj <= ti)) A[(r0 + ti + 16 * a) * CQLD + r0 + tj + 16 * b] -= acc[a][b];
      }
    }
    __syncthreads();
    CQA(3);
  }
  CQT(2);
  // Rcur = L^T D^-1 (upper); the second round's product with the first round's R is the caller's (one MFMA GEMM)
  for (int e = tid; e < CQW * CQW; e += 256) {
    const int i = e >> 7, j = e & 127;
    R[e] = (i <= j && i < w && j < w && dsc[j] > 0.0) ? A[j * CQLD + i] * dinv[j] : 0.0;
  }
  __syncthreads();
  CQT(3);
  // inverses of the 16 x 16 diagonal blocks, in place, two blocks per wave: lane i (< 16) holds row i of L11 and
  // forms COLUMN i of X11 = L11^-1:  x_ii = 1 / l_ii,  x_ri = -(sum_{k<r} l_rk x_ki) / l_rr  (operands of the other
  // rows by shuffles); every lane has read its row before any lane writes
  for (int kb = tid >> 6; kb < nbk; kb += 4) {
    const int k0 = kb * 16, i = lane & 15;
    double r[16], x[16], rinv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      r[j] = A[(k0 + i) * CQLD + k0 + j];
      rinv[j] = xd[k0 + j];
    }
#pragma unroll
    for (int ii = 0; ii < 16; ++ii) {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int k = 0; k < ii; ++k) {
        if (k & 1) s1 = fma(__shfl(r[k], ii, 64), x[k], s1);
        else s0 = fma(__shfl(r[k], ii, 64), x[k], s0);
      }
      x[ii] = (ii < i) ? 0.0 : (ii == i ? rinv[ii] : -(s0 + s1) * rinv[ii]);
    }
    if (lane < 16) {
#pragma unroll
      for (int ii = 0; ii < 16; ++ii)
        if (ii >= i) A[(k0 + ii) * CQLD + k0 + i] = x[ii];               // X11[ii][i], lower incl. diagonal
    }
  }
  __syncthreads();
  CQT(4);
  // X = L^-1 in place of the block-lower part, right-looking over the block columns J:  once X[J, 0 .. j0+16) is
  // final, every block row I > J takes  W[I, :] -= L[I, J] X[J, :]  (W: the running right-hand side of L X = I,
  // kept where L's consumed blocks were; L[., J] is saved in `tmp` first), in 7 x 8 register tiles
  for (int J = 0; J < nbk; ++J) {
    const int j0 = J * 16;
    if (J > 0) {
      // X[J, c] = X[J,J] W[J, c], c < j0
      // thread (r, cg) forms X[J][r][cg + 16 b], b < J: one entry of X[J,J] and J entries of W per step
      const int r = tid >> 4, cg = tid & 15;
      double xv[7];
#pragma unroll
      for (int b = 0; b < 7; ++b) xv[b] = 0.0;
#pragma unroll 4
      for (int t = 0; t < 16; ++t) {
        const double xdv = t <= r ? A[(j0 + r) * CQLD + j0 + t] : 0.0;
#pragma unroll
        for (int b = 0; b < 7; ++b)
          if (b < J) xv[b] = fma(xdv, A[(j0 + t) * CQLD + cg + 16 * b], xv[b]);
      }
      __syncthreads();
#pragma unroll
      for (int b = 0; b < 7; ++b)
        if (b < J) A[(j0 + r) * CQLD + cg + 16 * b] = xv[b];
    }
    const int r0 = j0 + 16, nt = (wp - r0) >> 4;      // block rows below
    if (nt <= 0) break;                               // uniform
    for (int e = tid; e < nt * 256; e += 256) tmp[e] = A[(r0 + (e >> 4)) * CQLD + j0 + (e & 15)];
    __syncthreads();
    {
      const int ti = tid >> 4, tj = tid & 15;
      double acc[7][8];
#pragma unroll
      for (int a = 0; a < 7; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = 0.0;
#pragma unroll 2
      for (int t = 0; t < 16; ++t) {
        // all 15 operands are loaded unconditionally (every address is inside the buffers; a guarded LDS load
        // becomes a masked block with its own wait) and masked afterwards
        double li[7], xj[8];
#pragma unroll
        for (int a = 0; a < 7; ++a) li[a] = tmp[(ti + 16 * a) * 16 + t];
#pragma unroll
        for (int b = 0; b < 8; ++b) xj[b] = A[(j0 + t) * CQLD + tj + 16 * b];
#pragma unroll
        for (int b = 0; b < 8; ++b)
          // X[J][t][c], c = tj + 16 b: columns of block J itself only up to the diagonal (c - j0 <= t); columns
          // behind block J are not written below
          if (b == J && tj > t) xj[b] = 0.0;
#pragma unroll
        for (int a = 0; a < 7; ++a)
#pragma unroll
          for (int b = 0; b < 8; ++b) acc[a][b] = fma(li[a], xj[b], acc[a][b]);
      }
#pragma unroll
      for (int a = 0; a < 7; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b)
          if (a < nt && b <= J) {
            double* dst = &A[(r0 + ti + 16 * a) * CQLD + tj + 16 * b];
            *dst = (b == J ? 0.0 : *dst) - acc[a][b];      // block J of these rows held L[., J] (now in tmp)
          }
    }
    __syncthreads();
  }
  CQT(5);
  // T = D X^T (upper; zero for failed / padded columns)
  for (int e = tid; e < CQW * CQW; e += 256) {
    const int i = e >> 7, j = e & 127;
    T[e] = (i <= j && j < w && dsc[j] > 0.0) ? dsc[i] * A[j * CQLD + i] : 0.0;
  }
  CQT(6);
  CQP();
  if (__syncthreads_or(bad ? 1 : 0) && tid == 0) atomicExch(flag, 1);
}
void launch_cholqr_wide(hipStream_t st, int w, const double* G, int ldg, double* T, double* R, int* flag) {
  const size_t lds = (size_t)(CQW * CQLD + 3 * CQW + 16 * 112) * sizeof(double);
  static bool attr_set = false;   // 147 KB of dynamic LDS: above the 64 KB default limit
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cholqr_wide_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(cholqr_wide_kernel, dim3(1), dim3(256), lds, st, w, G, ldg, T, R, flag);
}

// sel[i, jj] = evec[(c - 1 - jj), i]   (c x k, row-major): the k eigenvectors of the largest
// eigenvalues, as columns, from the row-major view of a column-major eigenvector matrix with
// ascending eigenvalues (row j of the view = eigenvector j).
__global__ void select_evecs_kernel(int c, int k, const double* __restrict__ evec, double* __restrict__ sel) {
  const size_t n = (size_t)c * k;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / k), jj = (int)(e % k);
    sel[e] = evec[(size_t)(c - 1 - jj) * c + i];
  }
}
void launch_select_evecs(hipStream_t st, int c, int k, const double* evec, double* sel) {
  const size_t n = (size_t)c * k;
  if (!n) return;
  int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(select_evecs_kernel, dim3(grid), dim3(256), 0, st, c, k, evec, sel);
}

// out[j, i] = sgn(j) * in[i, j]  for a k x k matrix; sgn(j) = +1 for j < k1, sneg otherwise
__global__ void transpose_sign_kernel(int k, int k1, double sneg, const double* __restrict__ in,
                                      double* __restrict__ out) {
  size_t n = (size_t)k * k;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n;
       e += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(e / k), i = (int)(e % k);
    out[e] = (j < k1 ? 1.0 : sneg) * in[(size_t)i * k + j];
  }
}
void launch_transpose_sign(hipStream_t st, int k, int k1, double sneg, const double* in, double* out) {
  size_t n = (size_t)k * k;
  int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(transpose_sign_kernel, dim3(grid), dim3(256), 0, st, k, k1, sneg, in, out);
}

// out[j, i] = in[i, j]  (rows x cols -> cols x rows, both row-major with their own leading dimensions)
__global__ void transpose_kernel(int rows, int cols, const double* __restrict__ in, int ldi,
                                 double* __restrict__ out, int ldo) {
  __shared__ double tile[32][33];
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: 8 rows of the tile per pass
  for (int r = ty; r < 32; r += 8)
    tile[r][tx] = (i0 + r < rows && j0 + tx < cols) ? in[(size_t)(i0 + r) * ldi + j0 + tx] : 0.0;
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (j0 + r < cols && i0 + tx < rows) out[(size_t)(j0 + r) * ldo + i0 + tx] = tile[tx][r];
}
void launch_transpose(hipStream_t st, int rows, int cols, const double* in, int ldi, double* out, int ldo) {
  if (rows <= 0 || cols <= 0) return;
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, st, rows, cols,
                     in, ldi, out, ldo);
}

// ---------------------------------------------------------------------------
// K5c: pivoted Cholesky in blocks (LAPACK dpstrf's scheme: inside a block the pivots are chosen from
// the lazily updated diagonal, the trailing matrix is updated once per block).
//
// The matrix is nr x nc, nc >= nr, row-major: its leading nr x nr part is symmetric positive
// semi-definite, columns nr.. are carried along (the factor's rows then hold R^-T times them: one
// factorisation of [H | B] gives chol(H) AND the triangular solve with B).  Nothing is swapped: row j of
// the factor belongs to pivot j and keeps the ORIGINAL column order, with exact zeros in the columns of
// earlier pivots.
//
// pchol_panel_kernel: ONE workgroup, one thread per column (CPT columns per thread beyond 1024); the
// thread keeps its entries of the block's rows in registers.  Per pivot: argmax of the current diagonal
// (wave shuffles + 16 LDS slots), the owner of the pivot column publishes its block entries, every
// thread forms its entry of the new row from row `p` of the trailing matrix.  Two barriers per pivot.
// pchol_trail_kernel:  A -= R_b^T R_b over all nr x nc entries (64 x 64 tiles).
// ---------------------------------------------------------------------------
// One pivot step (J = position in the block: a template parameter, so that the thread's block entries
// rb[.][J] are registers -- inside a loop, even a fully unrolled one, the array went to scratch memory).
template <int CPT, int NB, int J>
__device__ __forceinline__ void pchol_steps(const double* __restrict__ A, int ld, int nr, int nc, double tol, int kmax,
                                            int rank0, double* __restrict__ Rout, int ldr, int* __restrict__ done,
                                            double (&rb)[CPT][NB], double (&base)[CPT], double (&dots)[CPT],
                                            bool (&cand)[CPT], bool (&zero)[CPT], double& d0, int& made,
                                            bool& stopped, double* s_rp, double* s_val, int* s_idx) {
  if constexpr (J < NB) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (stopped) return;                      // uniform
    if (rank0 + J >= kmax) {
      stopped = true;
      return;
    }
    // pivot = largest remaining diagonal entry (lowest index on ties)
    double bv = -1.0;
    int bi = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
      const double v = base[q] - dots[q];
      if (cand[q] && v > bv) {
        bv = v;
        bi = tid + q * 1024;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double ov = __shfl_xor(bv, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (ov > bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    if (lane == 0) {
      s_val[wv] = bv;
      s_idx[wv] = bi;
    }
    __syncthreads();
    bv = s_val[0];
    bi = s_idx[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) {
      const double ov = s_val[w];
      const int oi = s_idx[w];
      if (ov > bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    const int p = bi;
    const double dp = bv;
    if (rank0 + J == 0) d0 = dp;
    if (!(dp > tol * d0) || !(dp > 0.0) || p >= nr) {     // uniform: every thread holds the same (p, dp)
      stopped = true;
      return;
    }
    if (tid == (p & 1023)) {
#pragma unroll
      for (int q = 0; q < CPT; ++q)
        if (q == (p >> 10)) {
#pragma unroll
          for (int i = 0; i < J; ++i) s_rp[i] = rb[q][i];
        }
    }
    __syncthreads();
    const double sq = sqrt(dp), inv = 1.0 / sq;
    const double* __restrict__ arow = A + (size_t)p * ld;
    double* __restrict__ rrow = Rout + (size_t)(rank0 + J) * ldr;
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
      const int t = tid + q * 1024;
      if (t < nc) {
        double acc = arow[t];
#pragma unroll
        for (int i = 0; i < J; ++i) acc = fma(-s_rp[i], rb[q][i], acc);
        double r = zero[q] ? 0.0 : acc * inv;
        if (t == p) {
          r = sq;
          cand[q] = false;
          zero[q] = true;
          done[t] = 1;
        }
        rb[q][J] = r;
        dots[q] = fma(r, r, dots[q]);
        rrow[t] = r;
      }
    }
    made = J + 1;
    pchol_steps<CPT, NB, J + 1>(A, ld, nr, nc, tol, kmax, rank0, Rout, ldr, done, rb, base, dots, cand, zero, d0,
                                made, stopped, s_rp, s_val, s_idx);
  }
}

template <int CPT, int NB>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void pchol_panel_kernel(
    const double* __restrict__ A, int ld, int nr, int nc, double tol, int kmax, PcholState* __restrict__ stt,
    double* __restrict__ Rout, int ldr, int* __restrict__ done) {
  if (stt->stop) return;                      // uniform
  __shared__ double s_rp[NB];
  __shared__ double s_val[16];
  __shared__ int s_idx[16];
  const int tid = threadIdx.x;
  const int rank0 = stt->rank;
  double d0 = stt->d0;
  double rb[CPT][NB], base[CPT], dots[CPT];
  bool cand[CPT], zero[CPT];
#pragma unroll
  for (int q = 0; q < CPT; ++q) {
    const int t = tid + q * 1024;
    const bool isdone = t < nr ? done[t] != 0 : false;
    base[q] = t < nr ? A[(size_t)t * ld + t] : 0.0;
    dots[q] = 0.0;
    cand[q] = t < nr && !isdone;
    zero[q] = isdone;
  }
  int made = 0;
  bool stopped = false;
  pchol_steps<CPT, NB, 0>(A, ld, nr, nc, tol, kmax, rank0, Rout, ldr, done, rb, base, dots, cand, zero, d0, made,
                          stopped, s_rp, s_val, s_idx);
  if (tid == 0) {
    stt->d0 = d0;
    stt->rank = rank0 + made;
    stt->nblk = made;
    if (stopped || rank0 + made >= kmax) stt->stop = 1;
  }
}

__global__ __launch_bounds__(256) void pchol_trail_kernel(double* __restrict__ A, int ld, int nr, int nc,
                                                          const PcholState* __restrict__ stt,
                                                          const double* __restrict__ Rall, int ldr) {
  if (stt->stop) return;                      // the factorisation ended with the last panel
  const int nb = stt->nblk;
  const double* __restrict__ Rb = Rall + (size_t)(stt->rank - nb) * ldr;
  __shared__ double sa[32][64], sb[32][64];
  const int i0 = blockIdx.y * 64, t0 = blockIdx.x * 64, tid = threadIdx.x;
  for (int e = tid; e < 32 * 64; e += 256) {
    const int j = e >> 6, x = e & 63;
    sa[j][x] = (j < nb && i0 + x < nr) ? Rb[(size_t)j * ldr + i0 + x] : 0.0;
    sb[j][x] = (j < nb && t0 + x < nc) ? Rb[(size_t)j * ldr + t0 + x] : 0.0;
  }
  __syncthreads();
  const int ti = (tid >> 4) * 4, tj = (tid & 15) * 4;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  for (int j = 0; j < nb; ++j) {
    double av[4], bw[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) av[a] = sa[j][ti + a];
#pragma unroll
    for (int b = 0; b < 4; ++b) bw[b] = sb[j][tj + b];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = fma(av[a], bw[b], acc[a][b]);
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = i0 + ti + a, t = t0 + tj + b;
      if (i < nr && t < nc) A[(size_t)i * ld + t] -= acc[a][b];
    }
}
int pchol_block(int nc) { return nc <= 1024 ? 32 : nc <= 2048 ? 16 : nc <= 4096 ? 8 : 0; }
void launch_pchol_panel(hipStream_t st, const double* A, int ld, int nr, int nc, double tol, int kmax,
                        PcholState* stt, double* Rout, int ldr, int* done) {
  if (nc <= 1024)
    hipLaunchKernelGGL((pchol_panel_kernel<1, 32>), dim3(1), dim3(1024), 0, st, A, ld, nr, nc, tol, kmax, stt,
                       Rout, ldr, done);
  else if (nc <= 2048)
    hipLaunchKernelGGL((pchol_panel_kernel<2, 16>), dim3(1), dim3(1024), 0, st, A, ld, nr, nc, tol, kmax, stt,
                       Rout, ldr, done);
  else
    hipLaunchKernelGGL((pchol_panel_kernel<4, 8>), dim3(1), dim3(1024), 0, st, A, ld, nr, nc, tol, kmax, stt,
                       Rout, ldr, done);
}
void launch_pchol_trail(hipStream_t st, double* A, int ld, int nr, int nc, const PcholState* stt,
                        const double* Rall, int ldr) {
  hipLaunchKernelGGL(pchol_trail_kernel, dim3((nc + 63) / 64, (nr + 63) / 64), dim3(256), 0, st, A, ld, nr, nc,
                     stt, Rall, ldr);
}

// ---------------------------------------------------------------------------
// K4s: the Z blocks of an ADI sweep in one launch.  Block j = sum_s coef[j][s] U_s  (U_s: the nslot solution
// panels, n x m each, `ustride` doubles apart; only the first nrows rows are used) goes straight into the
// factor (columns zc0 + j m ..), and the squared column norms of all blocks are accumulated per workgroup
// (partial[wg][j m + c]; sweep_norms_kernel sums them).  Round 2 issued three launches per block.
// coef is the host's replicated layout coef[(j nslot + s) m + c] (the same value for every c).
// ---------------------------------------------------------------------------
constexpr int SWC_ROWS = 64, SWC_MAXS = 16;
// The norms steer the ADI's stopping decisions, and with rank-sharded sweeps every rank takes them on its own from
// the same gathered panels: all sums below run in a FIXED order (no atomics), so that the ranks get the same bits.
__global__ __launch_bounds__(256) void sweep_combine_kernel(int nrows, int m, int nslot, int G,
                                                            const double* __restrict__ U, size_t ustride,
                                                            const double* __restrict__ coef,
                                                            double* __restrict__ Z, int zld, int zc0,
                                                            double* __restrict__ partial) {
  __shared__ double cs[SWC_MAXS * SWC_MAXS];
  __shared__ double acc[SWC_MAXS][256];
  const int tid = threadIdx.x;
  for (int e = tid; e < G * nslot; e += 256) cs[e] = coef[(size_t)e * m];
  __syncthreads();
  const int r0 = blockIdx.x * SWC_ROWS;
  const int cnt = min(SWC_ROWS, nrows - r0) * m;
  const int nthr = (256 / m) * m;             // working threads: a thread's column is the same for all of its elements
  double racc[SWC_MAXS];
#pragma unroll
  for (int j = 0; j < SWC_MAXS; ++j) racc[j] = 0.0;
  if (tid < nthr)
    for (int e = tid; e < cnt; e += nthr) {
      const int r = r0 + e / m, cidx = e % m;
      double u[SWC_MAXS];
#pragma unroll
      for (int sl = 0; sl < SWC_MAXS; ++sl)
        u[sl] = sl < nslot ? U[(size_t)sl * ustride + (size_t)r * m + cidx] : 0.0;
#pragma unroll
      for (int j = 0; j < SWC_MAXS; ++j) {
        if (j < G) {
          double v = 0.0;
#pragma unroll
          for (int sl = 0; sl < SWC_MAXS; ++sl) v = fma(sl < nslot ? cs[j * nslot + sl] : 0.0, u[sl], v);
          Z[(size_t)r * zld + zc0 + j * m + cidx] = v;
          racc[j] = fma(v, v, racc[j]);
        }
      }
    }
#pragma unroll
  for (int j = 0; j < SWC_MAXS; ++j)
    if (j < G) acc[j][tid] = racc[j];
  __syncthreads();
  for (int e = tid; e < G * m; e += 256) {
    const int j = e / m, cidx = e % m;
    double s = 0.0;
    for (int k = cidx; k < nthr; k += m) s += acc[j][k];
    partial[(size_t)blockIdx.x * G * m + e] = s;
  }
}
__global__ __launch_bounds__(256) void sweep_norms_kernel(int nwg, int gm, const double* __restrict__ partial,
                                                          double* __restrict__ out) {
  // 32 outputs per workgroup, the partial rows dealt to 8 row slices (a single thread per output walked all
  // ~400 rows one dependent load after the other: 55 us)
  __shared__ double red[8][32];
  const int col = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + col;
  double s0 = 0.0, s1 = 0.0;
  if (e < gm) {
    int w = sl;
    for (; w + 8 < nwg; w += 16) {
      s0 += partial[(size_t)w * gm + e];
      s1 += partial[(size_t)(w + 8) * gm + e];
    }
    if (w < nwg) s0 += partial[(size_t)w * gm + e];
  }
  red[sl][col] = s0 + s1;
  __syncthreads();
  if (sl == 0 && e < gm)
    out[e] = ((red[0][col] + red[1][col]) + (red[2][col] + red[3][col])) +
             ((red[4][col] + red[5][col]) + (red[6][col] + red[7][col]));
}
bool sweep_combine_ok(int m, int nslot, int G) { return nslot <= SWC_MAXS && G <= SWC_MAXS && m <= RICADI_MAX_M; }
size_t sweep_combine_partial_len(int nrows, int m, int G) {
  return (size_t)((nrows + SWC_ROWS - 1) / SWC_ROWS) * G * m;
}
void launch_sweep_combine(hipStream_t st, int nrows, int m, int nslot, int G, const double* U, size_t ustride,
                          const double* coef, double* Z, int zld, int zc0, double* partial, double* norms2) {
  const int nwg = (nrows + SWC_ROWS - 1) / SWC_ROWS;
  hipLaunchKernelGGL(sweep_combine_kernel, dim3(nwg), dim3(256), 0, st, nrows, m, nslot, G, U, ustride, coef, Z,
                     zld, zc0, partial);
  hipLaunchKernelGGL(sweep_norms_kernel, dim3((G * m + 31) / 32), dim3(256), 0, st, nwg, G * m, partial, norms2);
}

// coarse matrix combine: out = beta*E0 + alpha*EM + EJ  (dense k x k)
__global__ void combine3_kernel(size_t n, const double* a0, const double* a1, const double* a2,
                                double alpha, double beta, double* out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    out[i] = beta * a0[i] + alpha * a1[i] + a2[i];
}
void launch_combine3(hipStream_t st, size_t n, const double* a0, const double* a1,
                     const double* a2, double alpha, double beta, double* out) {
  int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(combine3_kernel, dim3(grid), dim3(256), 0, st, n, a0, a1, a2, alpha, beta,
                     out);
}

// identity matrix (for getrs against I)


// ---------------------------------------------------------------------------
// K2p: the pressure step of the SIMPLE cycle in ONE launch (16-column panels, 32 x 32 Schur blocks):
//     t   = J z_v + (S Y)_p e - r_p          (CSR rows of J over the panel z, of the prolongated operator over
//                                              the coarse correction e; r_p from the FP64 or the FP16-stored vector)
//     z_p = Shat_b^-1 t[rows_b]               (FP64 MFMA 16x16x4 on the FP32- / FP64-stored block inverse)
// with the epilogue of the Schur sweep it replaces (plain copy for the J^T product of the last velocity sweep,
// coarse prolongation, FP32 copy).  Round 2 issued three dependent launches here (pressure rows of r - (S Y) e,
// J product, Schur sweep: 10 + 25 + 9 us at 16 groups, ~20 us of latency floor at one group).  One workgroup of
// four waves per block: the 16-lane rows of all four waves form the block's 32 rows of t in two passes (index /
// value chunks by one coalesced load, DPP row broadcasts, 16 gathers in flight as in spmm_kernel_v2), t goes
// through LDS, waves 0 and 1 apply the inverse.
// ---------------------------------------------------------------------------
template <class T, class RT>
__global__ __launch_bounds__(256) void pressure_step_kernel(
    GroupTab gt, int nblocks, const int* __restrict__ bptr, const int* __restrict__ rows, GroupPtrsT<T> invs,
    const int* __restrict__ jrp, const int* __restrict__ jci, const double* __restrict__ jv,
    const double* __restrict__ z, size_t gsz,
    const int* __restrict__ syrp, const int* __restrict__ syci, GroupPtrs syv, const double* __restrict__ ec, size_t gse,
    const RT* __restrict__ rp_, size_t gsr, double* __restrict__ out, size_t gso, ProlongArgs pa) {
  __shared__ double tl[32][17];
  const int grp = gt.gid[blockIdx.z];
  const T* __restrict__ inv = invs.p[grp] + (size_t)blockIdx.x * 1024;
  z += (size_t)grp * gsz;
  rp_ += (size_t)grp * gsr;
  out += (size_t)grp * gso;
  const double* __restrict__ sval = syrp ? syv.p[grp] : nullptr;
  const double* __restrict__ ecg = ec ? ec + (size_t)grp * gse : nullptr;
  const int b0 = bptr[blockIdx.x], nb = bptr[blockIdx.x + 1] - b0;
  const int g = threadIdx.x & 15, rg = threadIdx.x >> 4;        // column, row group (16 of them)
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int il = rg + 16 * pass;
    const bool live = il < nb;
    const int prow = live ? rows[b0 + il] : 0;
    double acc = 0.0;
    {
      const int k0 = live ? jrp[prow] : 0, k1 = live ? jrp[prow + 1] : 0;
      int nch = (k1 - k0 + 15) >> 4;
      nch = max(nch, __shfl_xor(nch, 16, 64));
      nch = max(nch, __shfl_xor(nch, 32, 64));
      for (int ch = 0; ch < nch; ++ch) {
        const int k = k0 + ch * 16 + g;
        int myc = 0;
        double myv = 0.0;
        if (k < k1) {
          myc = jci[k];
          myv = jv[k];
        }
#define RICADI_PS_STEP(TT)                                  \
  {                                                         \
    const int c0 = bc16i<TT>(myc);                          \
    const double v0 = bc16d<TT>(myv);                       \
    acc = fma(v0, z[(size_t)c0 * 16 + g], acc);             \
  }
        RICADI_FOR16(RICADI_PS_STEP)
#undef RICADI_PS_STEP
      }
    }
    if (syrp) {
      const int k0 = live ? syrp[prow] : 0, k1 = live ? syrp[prow + 1] : 0;
      int nch = (k1 - k0 + 7) >> 3;
      nch = max(nch, __shfl_xor(nch, 16, 64));
      nch = max(nch, __shfl_xor(nch, 32, 64));
      for (int ch = 0; ch < nch; ++ch) {
        const int k = k0 + ch * 8 + g;
        int myc = 0;
        double myv = 0.0;
        if (g < 8 && k < k1) {
          myc = syci[k];
          myv = sval[k];
        }
#define RICADI_PS_STEP(TT)                                  \
  {                                                         \
    const int c0 = bc16i<TT>(myc);                          \
    const double v0 = bc16d<TT>(myv);                       \
    acc = fma(v0, ecg[(size_t)c0 * 16 + g], acc);           \
  }
        RICADI_PS_STEP(0) RICADI_PS_STEP(1) RICADI_PS_STEP(2) RICADI_PS_STEP(3)
        RICADI_PS_STEP(4) RICADI_PS_STEP(5) RICADI_PS_STEP(6) RICADI_PS_STEP(7)
#undef RICADI_PS_STEP
      }
    }
    const double rv = live ? (double)rp_[(size_t)prow * 16 + g] : 0.0;
    tl[il][g] = live ? acc - rv : 0.0;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  if (wave >= 2) return;
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4, t = wave;
  d4 acc4 = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kc = 0; kc < 2; ++kc) {
    double a4[4];
    load4(inv + (size_t)(16 * t + r) * 32 + kc * 16 + 4 * q, a4);
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)
      acc4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[s2], tl[kc * 16 + 4 * q + s2][r], acc4, 0, 0, 0);
  }
  const double* __restrict__ pec = pa.aggof ? pa.ec + (size_t)grp * pa.gse : nullptr;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int il = 16 * t + q + 4 * e;
    if (il < nb) {
      const int row = rows[b0 + il];
      double v = acc4[e];
      if (pa.out2) pa.out2[(size_t)grp * pa.gs2 + (size_t)row * 16 + r] = v;
      if (pec) v += pec[(size_t)pa.aggof[row] * 16 + r];
      if (!(pa.out32 && pa.only32)) out[(size_t)row * 16 + r] = v;
      if (pa.out32) pa.out32[(size_t)grp * pa.gs32 + (size_t)row * 16 + r] = (float)v;
    }
  }
}
template <class T>
static void pressure_step_impl(hipStream_t st, const GroupTab& gt, int nblocks, const int* bptr, const int* rows,
                               const GroupPtrsT<T>& inv, const int* jrp, const int* jci, const double* jv, const double* z,
                               size_t gsz, const int* syrp, const int* syci, const GroupPtrs& syv, const double* ec,
                               size_t gse, const double* rp_, const _Float16* rp16, size_t gsr, double* out, size_t gso,
                               const ProlongArgs& pa) {
  if (nblocks <= 0 || gt.ng <= 0) return;
  dim3 grid(nblocks, 1, gt.ng), block(256);
  if (rp16)
    hipLaunchKernelGGL((pressure_step_kernel<T, _Float16>), grid, block, 0, st, gt, nblocks, bptr, rows, inv, jrp, jci, jv, z,
                       gsz, syrp, syci, syv, ec, gse, rp16, gsr, out, gso, pa);
  else
    hipLaunchKernelGGL((pressure_step_kernel<T, double>), grid, block, 0, st, gt, nblocks, bptr, rows, inv, jrp, jci, jv, z,
                       gsz, syrp, syci, syv, ec, gse, rp_, gsr, out, gso, pa);
}
void launch_pressure_step_b(hipStream_t st, const GroupTab& gt, int nblocks, const int* bptr, const int* rows,
                            const GroupPtrsF& inv, const int* jrp, const int* jci, const double* jv, const double* z,
                            size_t gsz, const int* syrp, const int* syci, const GroupPtrs& syv, const double* ec, size_t gse,
                            const double* rp_, const _Float16* rp16, size_t gsr, double* out, size_t gso,
                            const ProlongArgs& pa) {
  pressure_step_impl(st, gt, nblocks, bptr, rows, inv, jrp, jci, jv, z, gsz, syrp, syci, syv, ec, gse, rp_, rp16, gsr, out,
                     gso, pa);
}
void launch_pressure_step_b(hipStream_t st, const GroupTab& gt, int nblocks, const int* bptr, const int* rows,
                            const GroupPtrs& inv, const int* jrp, const int* jci, const double* jv, const double* z,
                            size_t gsz, const int* syrp, const int* syci, const GroupPtrs& syv, const double* ec, size_t gse,
                            const double* rp_, const _Float16* rp16, size_t gsr, double* out, size_t gso,
                            const ProlongArgs& pa) {
  pressure_step_impl(st, gt, nblocks, bptr, rows, inv, jrp, jci, jv, z, gsz, syrp, syci, syv, ec, gse, rp_, rp16, gsr, out,
                     gso, pa);
}

// ---------------------------------------------------------------------------
// K3h: the last Arnoldi pass of the hot path (FP16-stored basis, 16 columns) WITH the Hessenberg / Givens update
// in the same launch -- one dependent launch per iteration less.  Every workgroup derives the normalisation
// 1 / h_{j+1,j} of its 16 columns itself from the (already reduced) Gram-Schmidt coefficients,
//     h_{j+1,j}^2 = ||w'||^2 - sum_i h2_i^2,   frozen columns (converged, or exact breakdown) -> 0,
// so nothing it needs comes from another workgroup of the launch; workgroup 0 of every group ALSO does what
// gmres_hess_kernel did (column of H through the stored rotations, new rotation, g, residual estimate into
// pinned host memory).  The residual estimates are double buffered (resid_in read by everybody, resid_out
// written by workgroup 0): a value that decides "frozen" must not change under the other workgroups' feet.
//   use_sum = 1: w is the vector BEFORE the first projection, coefficients h1 + h2 (cols_update_dots16<.., false>);
//   use_sum = 0: w has been projected once, coefficients h2.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cols_update16_hess_kernel(
    GroupTab gt, size_t nhalf, int nvec, const _Float16* __restrict__ basis, size_t vstride, size_t gsb,
    const double* __restrict__ h1, const double* __restrict__ h2, size_t gsh, int use_sum,
    const double* __restrict__ w, size_t gsw, double* __restrict__ out, size_t gso, _Float16* __restrict__ outf,
    size_t gsf, int j, int restart, double* __restrict__ H, double* __restrict__ cs, double* __restrict__ sn,
    double* __restrict__ g, const double* __restrict__ resid_in, double* __restrict__ resid_out,
    const double* __restrict__ bnorm, double tol, double* __restrict__ host_resid) {
  extern __shared__ double hl[];           // nvec x 16 coefficients, then 16 scales
  const int m = 16;
  const int grp = gt.gid[blockIdx.z];
  basis += (size_t)grp * gsb;
  h1 += (size_t)grp * gsh;
  h2 += (size_t)grp * gsh;
  w += (size_t)grp * gsw;
  if (out) out += (size_t)grp * gso;
  outf += (size_t)grp * gsf;
  double* scl = hl + nvec * m;
  const double tiny = 1e-300;
  for (int e = threadIdx.x; e < nvec * m; e += 256) hl[e] = use_sum ? h1[e] + h2[e] : h2[e];
  double hnext = 0.0;
  if (threadIdx.x < m) {
    const int c = threadIdx.x;
    double h2sq = 0.0;
    for (int i = 0; i < nvec; ++i) {
      const double b = h2[i * m + c];
      h2sq = fma(b, b, h2sq);
    }
    const double hn2 = h2[nvec * m + c] - h2sq;          // ||w'||^2 before the second projection, minus it
    hnext = hn2 > 0.0 ? sqrt(hn2) : 0.0;
    const double rprev = resid_in[(size_t)grp * m + c];  // |g_j|
    if (!(hnext > tiny) || rprev <= 0.01 * tol * bnorm[(size_t)grp * m + c]) hnext = 0.0;   // frozen column
    scl[c] = hnext > tiny ? 1.0 / hnext : 0.0;
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x < m) {
    // the Hessenberg column of this iteration (what gmres_hess_kernel did), one lane per panel column
    const int c = threadIdx.x;
    const size_t gq = (size_t)grp;
    double* Hc = H + gq * m * (restart + 1) * restart + (size_t)c * (restart + 1) * restart + (size_t)j * (restart + 1);
    double* csc = cs + gq * m * restart + (size_t)c * restart;
    double* snc = sn + gq * m * restart + (size_t)c * restart;
    double* gc = g + gq * m * (restart + 1) + (size_t)c * (restart + 1);
    const double gj = gc[j];
    double cur = h1[c] + h2[c];
    for (int i = 0; i < j; ++i) {
      const double nxt = h1[(i + 1) * m + c] + h2[(i + 1) * m + c];
      const double t = csc[i] * cur + snc[i] * nxt;
      const double u = -snc[i] * cur + csc[i] * nxt;
      Hc[i] = t;
      cur = u;
    }
    const double d = hypot(cur, hnext);
    double cj = 1.0, sj = 0.0;
    if (d > tiny) {
      cj = cur / d;
      sj = hnext / d;
    }
    csc[j] = cj;
    snc[j] = sj;
    Hc[j] = (d > tiny) ? d : 1.0;          // keep R non-singular for frozen columns
    Hc[j + 1] = 0.0;
    gc[j + 1] = (d > tiny) ? -sj * gj : 0.0;
    gc[j] = (d > tiny) ? cj * gj : 0.0;
    const double rnew = (d > tiny) ? fabs(sj * gj) : 0.0;
    resid_out[gq * m + c] = rnew;
    if (host_resid) host_resid[gq * m + c] = rnew;
  }
  for (size_t idx = blockIdx.x * (size_t)256 + threadIdx.x; idx < nhalf; idx += (size_t)gridDim.x * 256) {
    const size_t e = idx * 8;
    const int c0 = (int)(idx & 1) * 8;
    double a[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) a[t] = 0.0;
    const _Float16* v = basis + e;
    int i = 0;
    for (; i + 3 < nvec; i += 4) {
      half8_t x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const half8_t*>(v + (size_t)(i + u) * vstride);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 8; ++t) a[t] = fma(hl[(i + u) * m + c0 + t], (double)x[u][t], a[t]);
    }
    for (; i < nvec; ++i) {
      const half8_t x = *reinterpret_cast<const half8_t*>(v + (size_t)i * vstride);
#pragma unroll
      for (int t = 0; t < 8; ++t) a[t] = fma(hl[i * m + c0 + t], (double)x[t], a[t]);
    }
    const double2* wp = reinterpret_cast<const double2*>(w + e);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double2 ww = wp[t];
      a[2 * t] = (ww.x - a[2 * t]) * scl[c0 + 2 * t];
      a[2 * t + 1] = (ww.y - a[2 * t + 1]) * scl[c0 + 2 * t + 1];
    }
    half8_t f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f[t] = (_Float16)a[t];
      a[t] = (double)f[t];
    }
    *reinterpret_cast<half8_t*>(outf + e) = f;
    if (out) {
      double2* op = reinterpret_cast<double2*>(out + e);
#pragma unroll
      for (int t = 0; t < 4; ++t) op[t] = make_double2(a[2 * t], a[2 * t + 1]);
    }
  }
}
bool update_hess_fused_ok(int m, bool fp16_basis) {
  static const bool on = true;
  return on && fp16_basis && m == 16 && arnoldi16(4);
}
void launch_cols_update16_hess_b(hipStream_t st, const GroupTab& gt, int nrows, int nvec, const _Float16* basis,
                                 size_t vstride, size_t gsb, const double* h1, const double* h2, size_t gsh, int use_sum,
                                 const double* w, size_t gsw, double* out, size_t gso, _Float16* outf, size_t gsf, int j,
                                 int restart, double* H, double* cs, double* sn, double* g, const double* resid_in,
                                 double* resid_out, const double* bnorm, double tol, double* host_resid) {
  if (gt.ng <= 0) return;
  const size_t nhalf = (size_t)nrows * 2;
  const int grid = (int)std::min<size_t>((nhalf + 255) / 256, 8192);
  hipLaunchKernelGGL(cols_update16_hess_kernel, dim3(grid, 1, gt.ng), dim3(256), (size_t)(nvec * 16 + 16) * sizeof(double),
                     st, gt, nhalf, nvec, basis, vstride, gsb, h1, h2, gsh, use_sum, w, gsw, out, gso, outf, gsf, j, restart,
                     H, cs, sn, g, resid_in, resid_out, bnorm, tol, host_resid);
}

}  // namespace ricadi
