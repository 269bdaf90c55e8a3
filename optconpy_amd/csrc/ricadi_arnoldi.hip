// ricadi_arnoldi.hip -- K3: the Arnoldi passes of the lockstep GMRES on a low-precision-stored Krylov basis
// (dots, update+dots, update + Hessenberg) and the small per-column GMRES kernels.
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
// WT: storage type of the panel w (round 4: the operator's output of the hot path is an FP32 panel -- the basis it is
// orthogonalised against is FP16-stored, so its rounding of 6e-8 is far inside what the iteration already carries;
// the arithmetic stays FP64)
template <bool ATOMIC = false, class WT = double>
__global__ __launch_bounds__(256) void cols_dots16_kernel(
    GroupTab gt, int nrows, int nvec, const _Float16* __restrict__ basis, size_t vstride, size_t gsb,
    const WT* __restrict__ w, size_t gsw, int want_self, double* __restrict__ partial, size_t gsp) {
  __shared__ __attribute__((aligned(16))) double wl[DOT_ROWS * WLS];
  const int grp = gt.gid[blockIdx.z];
  basis += (size_t)grp * gsb;
  w += (size_t)grp * gsw;
  partial += (size_t)grp * gsp;
  const int r0 = blockIdx.x * DOT_ROWS;
  const int nr = min(DOT_ROWS, nrows - r0);
  if constexpr (sizeof(WT) == 4) {
    const float2* src = reinterpret_cast<const float2*>(w + (size_t)r0 * 16);
    float2 raw[DOT_ROWS * 8 / 256];
#pragma unroll
    for (int k = 0; k < DOT_ROWS * 8 / 256; ++k) {
      const int e = threadIdx.x + 256 * k;
      raw[k] = src[e < nr * 8 ? e : 0];
    }
#pragma unroll
    for (int k = 0; k < DOT_ROWS * 8 / 256; ++k) {
      const int e = threadIdx.x + 256 * k;
      *reinterpret_cast<double2*>(wl + (e >> 3) * WLS + (e & 7) * 2) =
          e < nr * 8 ? make_double2((double)raw[k].x, (double)raw[k].y) : make_double2(0.0, 0.0);
    }
  } else {
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
template <bool ATOMIC = false, bool STORE = true, class WT = double>
__global__ __launch_bounds__(256) void cols_update_dots16_kernel(
    GroupTab gt, int nrows, int nvec, const _Float16* __restrict__ basis, size_t vstride, size_t gsb,
    const double* __restrict__ h, size_t gsh, WT* __restrict__ w, size_t gsw,
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
    WT wv[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) wv[k] = w[base + (ok[k] ? e[k] : 0)];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const double v = ok[k] ? (double)wv[k] - sacc[k] : 0.0;
      if (STORE && ok[k]) w[base + e[k]] = (WT)v;
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
      hipLaunchKernelGGL((cols_dots16_kernel<false, double>), dim3(nblk, 1, gt.ng), dim3(256), 0, st, gt, nrows, nvec, basis,
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
        hipLaunchKernelGGL((cols_update_dots16_kernel<false, false, double>), dim3(nblk, 1, gt.ng), dim3(256),
                           (size_t)(DOT_ROWS * 18 + nvec * 16) * sizeof(double), st, gt, nrows, nvec, basis, vstride,
                           gsb, h, gsh, w, gsw, partial, gsp);
      else
        hipLaunchKernelGGL((cols_update_dots16_kernel<false, true, double>), dim3(nblk, 1, gt.ng), dim3(256),
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
template <class WT = double>
__global__ __launch_bounds__(256) void cols_update16_hess_kernel(
    GroupTab gt, size_t nhalf, int nvec, const _Float16* __restrict__ basis, size_t vstride, size_t gsb,
    const double* __restrict__ h1, const double* __restrict__ h2, size_t gsh, int use_sum,
    const WT* __restrict__ w, size_t gsw, double* __restrict__ out, size_t gso, _Float16* __restrict__ outf,
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
    if constexpr (sizeof(WT) == 4) {
      const float4* wp = reinterpret_cast<const float4*>(w + e);
      const float4 w0 = wp[0], w1 = wp[1];
      const float wf[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int t = 0; t < 8; ++t) a[t] = ((double)wf[t] - a[t]) * scl[c0 + t];
    } else {
      const double2* wp = reinterpret_cast<const double2*>(w + e);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double2 ww = wp[t];
        a[2 * t] = (ww.x - a[2 * t]) * scl[c0 + 2 * t];
        a[2 * t + 1] = (ww.y - a[2 * t + 1]) * scl[c0 + 2 * t + 1];
      }
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
                                 double* resid_out, const double* bnorm, double tol, double* host_resid,
                                 const float* w32) {
  if (gt.ng <= 0) return;
  const size_t nhalf = (size_t)nrows * 2;
  const int grid = (int)std::min<size_t>((nhalf + 255) / 256, 8192);
  if (w32)
    hipLaunchKernelGGL(cols_update16_hess_kernel<float>, dim3(grid, 1, gt.ng), dim3(256),
                       (size_t)(nvec * 16 + 16) * sizeof(double), st, gt, nhalf, nvec, basis, vstride, gsb, h1, h2, gsh,
                       use_sum, w32, gsw, out, gso, outf, gsf, j, restart, H, cs, sn, g, resid_in, resid_out, bnorm, tol,
                       host_resid);
  else
    hipLaunchKernelGGL(cols_update16_hess_kernel<double>, dim3(grid, 1, gt.ng), dim3(256),
                       (size_t)(nvec * 16 + 16) * sizeof(double), st, gt, nhalf, nvec, basis, vstride, gsb, h1, h2, gsh,
                       use_sum, w, gsw, out, gso, outf, gsf, j, restart, H, cs, sn, g, resid_in, resid_out, bnorm, tol,
                       host_resid);
}

// The first two Arnoldi passes on an FP32 panel w (16 columns, FP16-stored basis; the second pass in its
// "w kept" form: nothing is written back): same partial / reduce structure as the FP64-panel launches.
bool arnoldi16_w32_ok(int nvec_max) {
  return arnoldi16(1) && arnoldi16(2) && (size_t)(DOT_ROWS * 18 + nvec_max * 16) * sizeof(double) <= 48 * 1024;
}
void launch_cols_dots16_w32(hipStream_t st, const GroupTab& gt, int nrows, int nvec, const _Float16* basis,
                            size_t vstride, size_t gsb, const float* w32, size_t gsw, double* partial, size_t gsp,
                            double* out, size_t gso) {
  if (gt.ng <= 0 || nvec <= 0) return;
  const int nblk = dots_num_blocks(nrows), nout = nvec * 16;
  hipLaunchKernelGGL((cols_dots16_kernel<false, float>), dim3(nblk, 1, gt.ng), dim3(256), 0, st, gt, nrows, nvec, basis,
                     vstride, gsb, w32, gsw, 0, partial, gsp);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((nout + 15) / 16, 1, gt.ng), dim3(256), 0, st, gt, nblk, nout, partial,
                     gsp, out, gso, 0);
}
void launch_cols_update_dots16_w32(hipStream_t st, const GroupTab& gt, int nrows, int nvec, const _Float16* basis,
                                   size_t vstride, size_t gsb, const double* h, size_t gsh, float* w32, size_t gsw,
                                   double* partial, size_t gsp, double* out, size_t gso) {
  if (gt.ng <= 0) return;
  const int nblk = dots_num_blocks(nrows), nout = (nvec + 1) * 16;
  hipLaunchKernelGGL((cols_update_dots16_kernel<false, false, float>), dim3(nblk, 1, gt.ng), dim3(256),
                     (size_t)(DOT_ROWS * 18 + nvec * 16) * sizeof(double), st, gt, nrows, nvec, basis, vstride, gsb, h,
                     gsh, w32, gsw, partial, gsp);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((nout + 15) / 16, 1, gt.ng), dim3(256), 0, st, gt, nblk, nout, partial,
                     gsp, out, gso, 0);
}


}  // namespace ricadi
