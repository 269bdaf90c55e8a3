// ricadi_spmm.hip -- K1: the saddle operator on row-major panels (CSR, LDS-tiled, multi-shift) and the
// elementwise panel helpers (K4).
//
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
#include "ricadi_device.h"

namespace ricadi {

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
// K1b for LONG rows (the restriction P^T r of the preconditioner: one row per aggregate, 90 entries at cfg2, 390 at
// n = 5e5): ONE WAVE per row.  spmm_kernel_v2 gives a row to a 16-lane group, which walks its chunks one after the
// other -- index load, then 16 gathers, per chunk: two dependent rounds x 6 ... 24 chunks (17.5 us at cfg2, 201 us at
// n = 5e5 = 0.18 of the HBM roofline).  Here the four 16-lane groups of the wave take every fourth chunk, the next
// chunk's (index, value) pair travels while the current chunk's 16 gathers do, and the four partial sums meet in two
// cross-row shuffles (first form of round 4: 201 -> 155 us at n = 5e5, nothing at cfg2; the form below replaced it).
// Panels of 16 columns; no residual / scaling / low-rank terms.
// ---------------------------------------------------------------------------
typedef _Float16 rw_half8 __attribute__((ext_vector_type(8)));
template <int CTRL>
__device__ __forceinline__ double rw_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// Lane = (entry slot e = lane & 31, column half h = lane >> 5): per step a lane takes ONE entry and gathers the 8
// columns of its half of the x row with 16-byte loads (FP16 x: one load; FP64 x: four) -- 32 entries per wave step and
// one or four gather instructions, where the 16-lanes-per-row form issues 16 two-byte gathers per 16 entries and runs
// at the rate of the address path (20 cycles per gather instruction and CU at cfg2), not of the bytes.  The next step's
// (index, value) pair travels during the current gather; the 32 slots of a half are summed by four DPP exchanges and
// one cross-row shuffle.  m = 16 only.
template <class XT>
__global__ __launch_bounds__(256) void spmm_rowwave_kernel(
    GroupTab gt, int nrows, const int* __restrict__ rp, const int* __restrict__ ci, GroupPtrs vals,
    const XT* __restrict__ x, size_t gsx, double* __restrict__ y, size_t gsy) {
  const int grp = gt.gid[blockIdx.z];
  const double* __restrict__ val = vals.p[grp];
  x += (size_t)grp * gsx;
  y += (size_t)grp * gsy;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;                                  // wave-uniform
  const int lane = threadIdx.x & 63, e = lane & 31, h = lane >> 5;
  const int k0 = rp[row], k1 = rp[row + 1];
  const int klast = max(k1 - 1, k0);
  double acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] = 0.0;
  // batches of 4 steps (128 entries): the four (index, value) pairs are one load round, the four gathers the next --
  // a wave is a chain of dependent rounds (pointer -> pairs -> gathers -> ...), and with one step per round the launch
  // ran at that chain's latency whatever the instruction mix (18 us at cfg2 for 17 MB)
  constexpr int NB = 4;
  const int nbatch = (k1 - k0 + 32 * NB - 1) / (32 * NB);
  int cn[NB];
  double vn[NB];
  auto fetch = [&](int kb) {
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int k = kb + 32 * u + e;
      cn[u] = ci[min(k, klast)];
      const double v = val[min(k, klast)];
      vn[u] = k < k1 ? v : 0.0;
    }
  };
  if (nbatch > 0) fetch(k0);
  for (int bi = 0; bi < nbatch; ++bi) {
    int myc[NB];
    double myv[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      myc[u] = cn[u];
      myv[u] = vn[u];
    }
    if constexpr (sizeof(XT) == 2) {
      rw_half8 xv[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) xv[u] = *reinterpret_cast<const rw_half8*>(x + (size_t)myc[u] * 16 + h * 8);
      if (bi + 1 < nbatch) fetch(k0 + (bi + 1) * 32 * NB);         // uniform; travels with the gathers
#pragma unroll
      for (int u = 0; u < NB; ++u)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = fma(myv[u], (double)xv[u][t], acc[t]);
    } else {
      if (bi + 1 < nbatch) fetch(k0 + (bi + 1) * 32 * NB);
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const double2* xp = reinterpret_cast<const double2*>(x + (size_t)myc[u] * 16 + h * 8);
        const double2 a = xp[0], b = xp[1], c2 = xp[2], d = xp[3];
        acc[0] = fma(myv[u], a.x, acc[0]);
        acc[1] = fma(myv[u], a.y, acc[1]);
        acc[2] = fma(myv[u], b.x, acc[2]);
        acc[3] = fma(myv[u], b.y, acc[3]);
        acc[4] = fma(myv[u], c2.x, acc[4]);
        acc[5] = fma(myv[u], c2.y, acc[5]);
        acc[6] = fma(myv[u], d.x, acc[6]);
        acc[7] = fma(myv[u], d.y, acc[7]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    double v = acc[t];
    v += rw_dpp<0xB1>(v);     // quad_perm [1,0,3,2]
    v += rw_dpp<0x4E>(v);     // quad_perm [2,3,0,1]
    v += rw_dpp<0x141>(v);    // row_half_mirror
    v += rw_dpp<0x140>(v);    // row_mirror: all 16 lanes of the DPP row hold the row's sum
    v += __shfl_xor(v, 16, 64);
    acc[t] = v;
  }
  if (e == 0) {
    double2* o = reinterpret_cast<double2*>(y + (size_t)row * 16 + h * 8);
    o[0] = make_double2(acc[0], acc[1]);
    o[1] = make_double2(acc[2], acc[3]);
    o[2] = make_double2(acc[4], acc[5]);
    o[3] = make_double2(acc[6], acc[7]);
  }
}
bool spmm_rowwave_pays(int nrows, size_t nnz) { return nrows > 0 && nnz >= (size_t)32 * nrows; }
void launch_spmm_rowwave(hipStream_t st, const GroupTab& gt, int nrows, const int* rp, const int* ci,
                         const GroupPtrs& vals, const double* x, const _Float16* x16, size_t gsx, double* y, size_t gsy,
                         int m) {
  if (nrows <= 0 || m != 16 || gt.ng <= 0) return;
  dim3 grid((nrows + 3) / 4, 1, gt.ng), block(256);
  if (x16)
    hipLaunchKernelGGL((spmm_rowwave_kernel<_Float16>), grid, block, 0, st, gt, nrows, rp, ci, vals, x16, gsx, y, gsy);
  else
    hipLaunchKernelGGL((spmm_rowwave_kernel<double>), grid, block, 0, st, gt, nrows, rp, ci, vals, x, gsx, y, gsy);
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
    size_t gsr, double alpha, double beta_r, int m, int max_cols, LowRankArgs lr, float* __restrict__ y32) {
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
  if (y32) y32 += (size_t)grp * gsy;          // the product as an FP32 panel instead (same layout; Arnoldi reads it)
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
  if (sizeof(XT) == 4 && m == 16) {
    // FP32 rows of 16 columns are 64 B: a lane takes FOUR columns (16-byte load), a 16-lane group four rows per load
    // -- a quarter of the gather instructions of the one-column form.  (That form made the FP32-input kernel 10 %
    // slower than the FP64-input one -- same number of row requests, nothing gained from the smaller rows; two columns
    // per lane: 80.8 -> 65.5 us per 16-group launch at cfg2.)
    constexpr int XQ = 3;                      // 16 groups x 4 rows x 3 = 192 tile rows per pass
    const int sub = g >> 2, c4 = (g & 3) * 4;
    for (int jb = 0; jb < max_cols; jb += 64 * XQ) {
      int cidx[XQ];
      float4 xv[XQ];
#pragma unroll
      for (int t = 0; t < XQ; ++t) {
        const int j = jb + 4 * gq + sub + 64 * t;
        cidx[t] = (j < max_cols) ? bcols[j] : -1;
      }
#pragma unroll
      for (int t = 0; t < XQ; ++t)
        xv[t] = (cidx[t] >= 0) ? *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(x) + (size_t)cidx[t] * 16 + c4)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int t = 0; t < XQ; ++t) {
        const int j = jb + 4 * gq + sub + 64 * t;
        if (cidx[t] >= 0) {
          double2* d = reinterpret_cast<double2*>(xs + j * 16 + c4);
          d[0] = make_double2((double)xv[t].x, (double)xv[t].y);
          d[1] = make_double2((double)xv[t].z, (double)xv[t].w);
        }
      }
    }
  } else
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
    for (int ch = 0; ch < NCH; ++ch) {
      myl[rr][ch] *= (int)rowbytes;
      // pinned HERE: sunk by the optimiser into the branches below, the multiply would sit right in front of the
      // hand-written DPP instruction that reads it (round 4 saw exactly that in a variant of this kernel: all
      // columns but the first came out wrong)
      asm volatile("" : "+v"(myl[rr][ch]));
    }
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
        if (y32) y32[(size_t)row * ldy + cc] = (float)out;
        else y[(size_t)row * ldy + cc] = out;
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
                     vals, x, ldx, gsx, y, ldy, gsy, r, ldr, gsr, alpha, beta_r, m, max_cols, lr, (float*)nullptr)
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
                             size_t gsx, double* y, int ldy, size_t gsy, double alpha, int m, int max_cols, float* y32) {
  if (nblk <= 0 || gt.ng <= 0) return;
  const dim3 grid(nblk, 1, gt.ng), block(256);
  const size_t lds = spmm_blocked_lds_bytes(m, max_cols, 0);
  hipLaunchKernelGGL((spmm_blocked_kernel<false, false, float>), grid, block, lds, st, rows2, rp2, cols2, lidx, gt,
                     vals, x, ldx, gsx, y, ldy, gsy, (const double*)nullptr, 0, (size_t)0, alpha, 0.0, m, max_cols,
                     LowRankArgs(), y32);
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
template <bool HAS_R, class XT = double, bool F4 = false>
__global__ __launch_bounds__(256) void spmm_blocked_ms_kernel(
    const int* __restrict__ rows2, const int* __restrict__ rp2, const int* __restrict__ cols2,
    const uint16_t* __restrict__ lidx, GroupTab gt, GroupCoefs cf,
    const double* __restrict__ vAJ, const double* __restrict__ vE,
    const XT* __restrict__ x, int ldx, size_t gsx, double* __restrict__ y, int ldy, size_t gsy,
    const double* __restrict__ r, int ldr, size_t gsr, double alpha, double beta_r, int m,
    int max_cols, float* __restrict__ y32) {
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
  if constexpr (!F4) {
#pragma unroll
    for (int pp = 0; pp < XP; ++pp)
#pragma unroll
      for (int t = 0; t < XJ; ++t) {
        const int j = pp * 16 * XJ + gq + 16 * t;
        const int ci = (j < max_cols) ? bcols[j] : -1;
        xoff[pp][t] = ci >= 0 ? (ci * ldx + gc) * (int)sizeof(XT) : -1;
      }
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
  // FP32 rows of 16 columns (64 B): four columns per lane, 64 tile rows per pass of the workgroup -- three 16-byte
  // loads per thread and group instead of ten 4-byte ones (the per-group kernel gained 27 % from the same change)
  constexpr int XQ = 3;
  constexpr bool f4 = F4;                    // launcher: XT = float, m = ldx = 16
  const int j4 = threadIdx.x >> 2, c4 = (threadIdx.x & 3) * 4;
  int xoff4[XQ];
  float4 xv4[XQ];
  if constexpr (f4) {
#pragma unroll
    for (int t = 0; t < XQ; ++t) {
      const int j = j4 + 64 * t;
      const int ci = (j < max_cols) ? bcols[j] : -1;
      xoff4[t] = ci >= 0 ? (ci * 16 + c4) * 4 : -1;
    }
  }
  auto fetch = [&](int grp) {
    const char* __restrict__ xg = reinterpret_cast<const char*>(x + (size_t)grp * gsx);
    if constexpr (f4) {
#pragma unroll
      for (int t = 0; t < XQ; ++t)
        xv4[t] = (xoff4[t] >= 0) ? *reinterpret_cast<const float4*>(xg + (unsigned)xoff4[t]) : make_float4(0.f, 0.f, 0.f, 0.f);
      return;
    }
#pragma unroll
    for (int pp = 0; pp < XP; ++pp)
#pragma unroll
      for (int t = 0; t < XJ; ++t)
        xv[pp][t] = (xoff[pp][t] >= 0) ? *reinterpret_cast<const XT*>(xg + (unsigned)xoff[pp][t]) : (XT)0;
  };
  auto stash = [&](int buf) {
    double* __restrict__ tile = xs + (size_t)buf * max_cols * 16;
    if constexpr (f4) {
#pragma unroll
      for (int t = 0; t < XQ; ++t) {
        const int j = j4 + 64 * t;
        if (j < max_cols) {
          double2* d = reinterpret_cast<double2*>(tile + j * 16 + c4);
          d[0] = make_double2((double)xv4[t].x, (double)xv4[t].y);
          d[1] = make_double2((double)xv4[t].z, (double)xv4[t].w);
        }
      }
      return;
    }
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
        if (y32) y32[(size_t)grp * gsy + (size_t)row * ldy + g] = (float)out;
        else yg[(size_t)row * ldy + g] = out;
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
                                int ldy, size_t gsy, double alpha, int m, int max_cols, float* y32) {
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
  if (m == 16 && ldx == 16 && max_cols <= 192)
    hipLaunchKernelGGL((spmm_blocked_ms_kernel<false, float, true>), grid, block, spmm_blocked_ms_lds_bytes(max_cols), st,
                       rows2, rp2, cols2, lidx, gt, cf, vAJ, vE, x, ldx, gsx, y, ldy, gsy, (const double*)nullptr, 0,
                       (size_t)0, alpha, 0.0, m, max_cols, y32);
  else
    hipLaunchKernelGGL((spmm_blocked_ms_kernel<false, float, false>), grid, block, spmm_blocked_ms_lds_bytes(max_cols), st,
                       rows2, rp2, cols2, lidx, gt, cf, vAJ, vE, x, ldx, gsx, y, ldy, gsy, (const double*)nullptr, 0,
                       (size_t)0, alpha, 0.0, m, max_cols, y32);
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
                       cf, vAJ, vE, x, ldx, gsx, y, ldy, gsy, r, ldr, gsr, alpha, beta_r, m, max_cols, (float*)nullptr);
  else
    hipLaunchKernelGGL((spmm_blocked_ms_kernel<false>), grid, block, lds, st, rows2, rp2, cols2, lidx, gt,
                       cf, vAJ, vE, x, ldx, gsx, y, ldy, gsy, r, ldr, gsr, alpha, beta_r, m, max_cols, (float*)nullptr);
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


}  // namespace ricadi
