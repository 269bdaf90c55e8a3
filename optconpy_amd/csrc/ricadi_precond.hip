// ricadi_precond.hip -- K2: the multilevel preconditioner's sweeps (block-Jacobi, rectangular and two-term forms,
// fused pressure step), the coarse apply and the batched block Gauss-Jordan inverse of the coarse matrices.
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
typedef float f4v __attribute__((ext_vector_type(4)));      // accumulator of v_mfma_f32_16x16x4_f32
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
// BF16 bit patterns (uint16_t): widening is a 16-bit shift into the FP32 encoding
__device__ __forceinline__ double bf16_to_f64(unsigned bits16) { return (double)__uint_as_float(bits16 << 16); }
template <>
struct Raw4<uint16_t> {
  typedef uint2 type;
  static __device__ __forceinline__ uint2 load(const uint16_t* p) { return *reinterpret_cast<const uint2*>(p); }
  static __device__ __forceinline__ void unpack(const uint2& u, double (&a)[4]) {
    a[0] = bf16_to_f64(u.x & 0xffffu);
    a[1] = bf16_to_f64(u.x >> 16);
    a[2] = bf16_to_f64(u.y & 0xffffu);
    a[3] = bf16_to_f64(u.y >> 16);
  }
};
__device__ __forceinline__ void load4(const uint16_t* p, double (&a)[4]) {
  Raw4<uint16_t>::unpack(Raw4<uint16_t>::load(p), a);
}
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
// K2, the two velocity sweeps of the folded cycle for THE hot shape (32-row blocks, 16-column panels), round 4.
// The generic kernels above spend their time in dependent load rounds, not in bytes: block pointers -> row / input
// lists -> aggregate map -> gathers -> rows to update -> stores is five round trips per wave at 2-4 waves per SIMD
// (with the FP32 intermediate the last sweep moved 40 % fewer bytes in the same 738 us at n = 5e5).  Here
//   * every index of a block comes out of ONE fixed-stride record (ricadi_ctx::sw_meta, layout at
//     ProlongArgs::bmeta: lists padded with their last entry, the aggregate of every row stored with it);
//   * the gathers, the matrix tiles AND the rows the wave updates are requested together (the old rows do not
//     depend on the products);
//   * addresses are 32-bit byte offsets from uniform bases (one VGPR per load in flight instead of two), no column
//     loop, no column masks.
// So a wave needs two rounds (record, data) before its MFMAs.
// ---------------------------------------------------------------------------
template <class V>
__device__ __forceinline__ V ld_off(const V* base, unsigned byteoff) {
  return *reinterpret_cast<const V*>(reinterpret_cast<const char*>(base) + byteoff);
}
template <int KS, class T, bool OLD32>
__global__ __launch_bounds__(256) void block_rect32_kernel(
    GroupTab gt, int nblocks, const int* __restrict__ meta, int mstride, int in_off, GroupPtrsT<T> mats,
    const double* __restrict__ in, size_t gsi, double* __restrict__ out, size_t gso, int subtract, ProlongArgs pa) {
  const int grp = gt.gid[blockIdx.z];
  const T* __restrict__ mat = mats.p[grp];
  in += (size_t)grp * gsi;
  out += (size_t)grp * gso;
  const double* __restrict__ ec = pa.aggof ? pa.ec + (size_t)grp * pa.gse : nullptr;
  float* __restrict__ o32 = pa.out32 ? pa.out32 + (size_t)grp * pa.gs32 : nullptr;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  if (wave >= nblocks) {
    // surplus waves: coarse-level prolongation of the rows outside the blocks
    const int e0 = (wave - nblocks) * 32;
    for (int rr = e0 + q; rr < min(e0 + 32, pa.nextra); rr += 4) {
      const int row = pa.row0 + rr;
      const double v = out[(size_t)row * 16 + r] + ec[(size_t)pa.aggof[row] * 16 + r];
      out[(size_t)row * 16 + r] = v;
      if (o32) o32[(size_t)row * 16 + r] = (float)v;
    }
    return;
  }
  constexpr int NK = KS / 16;
  const int* __restrict__ mt = meta + (size_t)wave * mstride;
  const T* __restrict__ Gi = mat + (size_t)wave * 32 * KS;
  // round 1: the record
  const int nb = mt[0], ni = mt[1];
  int xrow[NK][4], orow[2][4], oagg[2][4];
#pragma unroll
  for (int kc = 0; kc < NK; ++kc)
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) xrow[kc][s2] = mt[in_off + kc * 16 + 4 * q + s2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      orow[t][e] = mt[4 + 16 * t + q + 4 * e];
      oagg[t][e] = mt[36 + 16 * t + q + 4 * e];
    }
  // round 2: gathers, tiles, old rows, coarse part -- raw values, converted behind the barrier
  const double* __restrict__ ecp = ec ? ec : in;         // a readable dummy when there is no coarse part
  double xb[NK][4], ecv[2][4], oldv[2][4];
  float oldf[2][4];
  typename Raw4<T>::type graw[2][NK];
#pragma unroll
  for (int kc = 0; kc < NK; ++kc)
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) xb[kc][s2] = ld_off(in, (unsigned)(xrow[kc][s2] * 16 + r) * 8u);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kc = 0; kc < NK; ++kc) graw[t][kc] = Raw4<T>::load(Gi + (16 * t + r) * KS + kc * 16 + 4 * q);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (OLD32) oldf[t][e] = ld_off(o32, (unsigned)(orow[t][e] * 16 + r) * 4u);
      else oldv[t][e] = ld_off(out, (unsigned)(orow[t][e] * 16 + r) * 8u);
      ecv[t][e] = ld_off(ecp, (unsigned)(oagg[t][e] * 16 + r) * 8u);
    }
  // pinned: the optimiser otherwise sinks the gathers (and their index loads) into the conditional chunks below,
  // behind the barrier -- one more dependent round per chunk
#pragma unroll
  for (int kc = 0; kc < NK; ++kc)
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) asm volatile("" : "+v"(xb[kc][s2]));
  __builtin_amdgcn_sched_barrier(0);
  d4 acc[2];
  acc[0] = acc[1] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kc = 0; kc < NK; ++kc) {
    if (kc * 16 >= ni) break;                 // wave-uniform: chunks beyond the block's inputs
    double xm[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) xm[s2] = (kc * 16 + 4 * q + s2 < ni) ? xb[kc][s2] : 0.0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      double a4[4];
      Raw4<T>::unpack(graw[t][kc], a4);
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[0], xm[0], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[1], xm[1], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[2], xm[2], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[3], xm[3], acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int il = 16 * t + q + 4 * e;
      if (il < nb) {
        const unsigned at = (unsigned)(orow[t][e] * 16 + r);
        const double old = OLD32 ? (double)oldf[t][e] : oldv[t][e];
        double v = subtract ? old - acc[t][e] : acc[t][e];
        if (pa.out2) pa.out2[(size_t)grp * pa.gs2 + at] = v;        // the result before the coarse part
        if (ec) v += ecv[t][e];
        if (!(o32 && pa.only32)) out[at] = v;
        if (o32) o32[at] = (float)v;
      }
    }
}

// first sweep:  out[rows_b] = M1_b * in1[rows_b] - M2_b * in2[list2_b]   (block_apply2_kernel for the hot shape)
template <int K2, class T, bool H1>
__global__ __launch_bounds__(256) void block_two32_kernel(
    GroupTab gt, int nblocks, const int* __restrict__ meta, int mstride, int in_off, GroupPtrsT<T> m1s, Seg2 s1,
    GroupPtrsT<T> m2s, Seg2 s2, double* __restrict__ out, size_t gso, ProlongArgs pa) {
  const int grp = gt.gid[blockIdx.z];
  out += (size_t)grp * gso;
  float* __restrict__ o32 = pa.out32 ? pa.out32 + (size_t)grp * pa.gs32 : nullptr;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wave >= nblocks) return;
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  constexpr int N2 = K2 / 16;
  const T* __restrict__ M1 = m1s.p[grp] + (size_t)wave * 32 * 32;
  const T* __restrict__ M2 = m2s.p[grp] + (size_t)wave * 32 * K2;
  const double* __restrict__ in1 = s1.in ? s1.in + (size_t)grp * s1.gs : nullptr;
  const _Float16* __restrict__ in1h = s1.in16 ? s1.in16 + (size_t)grp * s1.gs : nullptr;
  const double* __restrict__ in2 = s2.in + (size_t)grp * s2.gs;
  const int* __restrict__ mt = meta + (size_t)wave * mstride;
  // round 1: the record (lists padded with valid rows: no condition on any address)
  const int nb = mt[0], ni = mt[2];
  int r1[2][4], r2[N2][4], orow[2][4];
#pragma unroll
  for (int kc = 0; kc < 2; ++kc)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) r1[kc][s4] = mt[4 + kc * 16 + 4 * q + s4];
#pragma unroll
  for (int kc = 0; kc < N2; ++kc)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) r2[kc][s4] = mt[in_off + kc * 16 + 4 * q + s4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) orow[t][e] = mt[4 + 16 * t + q + 4 * e];
  // round 2: both input panels and the four / six matrix tiles, raw
  _Float16 h1[2][4];
  double x1[2][4], x2[N2][4];
  typename Raw4<T>::type a1[2][2], a2[2][N2];
#pragma unroll
  for (int kc = 0; kc < 2; ++kc)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      if (H1) h1[kc][s4] = ld_off(in1h, (unsigned)(r1[kc][s4] * 16 + r) * 2u);
      else x1[kc][s4] = ld_off(in1, (unsigned)(r1[kc][s4] * 16 + r) * 8u);
    }
#pragma unroll
  for (int kc = 0; kc < N2; ++kc)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) x2[kc][s4] = ld_off(in2, (unsigned)(r2[kc][s4] * 16 + r) * 8u);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) a1[t][kc] = Raw4<T>::load(M1 + (16 * t + r) * 32 + kc * 16 + 4 * q);
#pragma unroll
    for (int kc = 0; kc < N2; ++kc) a2[t][kc] = Raw4<T>::load(M2 + (16 * t + r) * K2 + kc * 16 + 4 * q);
  }
  // pinned (see block_rect32_kernel): the second segment's gathers are used under a wave-uniform condition
#pragma unroll
  for (int kc = 0; kc < N2; ++kc)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) asm volatile("" : "+v"(x2[kc][s4]));
  __builtin_amdgcn_sched_barrier(0);
  d4 acc[2];
  acc[0] = acc[1] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kc = 0; kc < 2; ++kc) {
    double xm[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const double v = H1 ? (double)h1[kc][s4] : x1[kc][s4];
      xm[s4] = (kc * 16 + 4 * q + s4 < nb) ? v : 0.0;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      double a4[4];
      Raw4<T>::unpack(a1[t][kc], a4);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[s4], xm[s4], acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int kc = 0; kc < N2; ++kc) {
    if (kc * 16 >= ni) break;                 // wave-uniform
    double xm[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) xm[s4] = (kc * 16 + 4 * q + s4 < ni) ? -x2[kc][s4] : 0.0;     // minus: one accumulator
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      double a4[4];
      Raw4<T>::unpack(a2[t][kc], a4);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[s4], xm[s4], acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int il = 16 * t + q + 4 * e;
      if (il < nb) {
        const unsigned at = (unsigned)(orow[t][e] * 16 + r);
        const double v = acc[t][e];
        if (pa.out2) pa.out2[(size_t)grp * pa.gs2 + at] = v;
        if (!(o32 && pa.only32)) out[at] = v;
        if (o32) o32[at] = (float)v;
      }
    }
}

// The first sweep on the FP32 matrix cores (v_mfma_f32_16x16x4_f32; BF16-stored blocks widened by a shift, the two
// input panels rounded to FP32, FP32 accumulation over the block's <= 96 terms): block_two32_kernel above does 1.30
// GFLOP per 16-group launch at cfg2 in 34 us = 0.49 of the FP64 matrix peak beside its 0.46 of HBM.  The arithmetic
// was mirrored on scipy (tools/schur_lab.py `sa+b16+m32`: iteration counts identical).  As the coarse apply's FP32
// form: WRITTEN IN THE LAST SESSION OF ROUND 4 WITHOUT GPU-MINUTES LEFT, never run on the device, off unless
// RICADI_SWEEP32=1.  Differences from the kernel above: operand and accumulator types, and
// the C/D map of the FP32 form -- row = 4 (l >> 4) + reg instead of (l >> 4) + 4 reg -- in the output rows.
template <int K2, bool H1>
__global__ __launch_bounds__(256) void block_two32_f32mfma_kernel(
    GroupTab gt, int nblocks, const int* __restrict__ meta, int mstride, int in_off, GroupPtrsT<uint16_t> m1s, Seg2 s1,
    GroupPtrsT<uint16_t> m2s, Seg2 s2, double* __restrict__ out, size_t gso, ProlongArgs pa) {
  const int grp = gt.gid[blockIdx.z];
  out += (size_t)grp * gso;
  float* __restrict__ o32 = pa.out32 ? pa.out32 + (size_t)grp * pa.gs32 : nullptr;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wave >= nblocks) return;
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  constexpr int N2 = K2 / 16;
  const uint16_t* __restrict__ M1 = m1s.p[grp] + (size_t)wave * 32 * 32;
  const uint16_t* __restrict__ M2 = m2s.p[grp] + (size_t)wave * 32 * K2;
  const double* __restrict__ in1 = s1.in ? s1.in + (size_t)grp * s1.gs : nullptr;
  const _Float16* __restrict__ in1h = s1.in16 ? s1.in16 + (size_t)grp * s1.gs : nullptr;
  const double* __restrict__ in2 = s2.in + (size_t)grp * s2.gs;
  const int* __restrict__ mt = meta + (size_t)wave * mstride;
  const int nb = mt[0], ni = mt[2];
  int r1[2][4], r2[N2][4], orow[2][4];
#pragma unroll
  for (int kc = 0; kc < 2; ++kc)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) r1[kc][s4] = mt[4 + kc * 16 + 4 * q + s4];
#pragma unroll
  for (int kc = 0; kc < N2; ++kc)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) r2[kc][s4] = mt[in_off + kc * 16 + 4 * q + s4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) orow[t][e] = mt[4 + 16 * t + 4 * q + e];      // FP32 C/D map
  _Float16 h1[2][4];
  double x1[2][4], x2[N2][4];
  uint2 a1[2][2], a2[2][N2];
#pragma unroll
  for (int kc = 0; kc < 2; ++kc)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      if (H1) h1[kc][s4] = ld_off(in1h, (unsigned)(r1[kc][s4] * 16 + r) * 2u);
      else x1[kc][s4] = ld_off(in1, (unsigned)(r1[kc][s4] * 16 + r) * 8u);
    }
#pragma unroll
  for (int kc = 0; kc < N2; ++kc)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) x2[kc][s4] = ld_off(in2, (unsigned)(r2[kc][s4] * 16 + r) * 8u);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) a1[t][kc] = Raw4<uint16_t>::load(M1 + (16 * t + r) * 32 + kc * 16 + 4 * q);
#pragma unroll
    for (int kc = 0; kc < N2; ++kc) a2[t][kc] = Raw4<uint16_t>::load(M2 + (16 * t + r) * K2 + kc * 16 + 4 * q);
  }
#pragma unroll
  for (int kc = 0; kc < N2; ++kc)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) asm volatile("" : "+v"(x2[kc][s4]));
  __builtin_amdgcn_sched_barrier(0);
  auto widen = [](const uint2& u, float (&a)[4]) {
    a[0] = __uint_as_float(u.x << 16);
    a[1] = __uint_as_float(u.x & 0xffff0000u);
    a[2] = __uint_as_float(u.y << 16);
    a[3] = __uint_as_float(u.y & 0xffff0000u);
  };
  f4v acc[2];
  acc[0] = acc[1] = (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kc = 0; kc < 2; ++kc) {
    float xm[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const float v = H1 ? (float)h1[kc][s4] : (float)x1[kc][s4];
      xm[s4] = (kc * 16 + 4 * q + s4 < nb) ? v : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float a4[4];
      widen(a1[t][kc], a4);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s4], xm[s4], acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int kc = 0; kc < N2; ++kc) {
    if (kc * 16 >= ni) break;                 // wave-uniform
    float xm[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) xm[s4] = (kc * 16 + 4 * q + s4 < ni) ? -(float)x2[kc][s4] : 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float a4[4];
      widen(a2[t][kc], a4);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s4], xm[s4], acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int il = 16 * t + 4 * q + e;
      if (il < nb) {
        const unsigned at = (unsigned)(orow[t][e] * 16 + r);
        const double v = (double)acc[t][e];
        if (pa.out2) pa.out2[(size_t)grp * pa.gs2 + at] = v;
        if (!(o32 && pa.only32)) out[at] = v;
        if (o32) o32[at] = acc[t][e];
      }
    }
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
    if (pa.old32) {               // uniform: the FP32 intermediate of the cycle (raw loads first, conversion behind them)
      const float* __restrict__ o32 = pa.out32 + (size_t)grp * pa.gs32;
      float oldf[NT][4];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          oldf[t][e] = o32[(size_t)orow[t][e] * ldo + colx];
          ecv[t][e] = ecp[(size_t)oagg[t][e] * m + colx];
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) oldv[t][e] = (double)oldf[t][e];
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          oldv[t][e] = out[(size_t)orow[t][e] * ldo + colx];
          ecv[t][e] = ecp[(size_t)oagg[t][e] * m + colx];
        }
      __builtin_amdgcn_sched_barrier(0);
    }
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
  if (pa.bmeta && bs == 32 && m == 16 && ldi == 16 && ldo == 16 && (ks == 32 || ks == 64) &&
      std::max(gsi, gso) * 8 < ((size_t)1 << 32) && !(pa.old32 && !pa.out32)) {
#define RICADI_RECT32(K, O)                                                                                  \
  hipLaunchKernelGGL((block_rect32_kernel<K, T, O>), grid, block, 0, st, gt, nblocks, pa.bmeta, pa.bm_stride, \
                     pa.bm_in, mats, in, gsi, out, gso, subtract, pa)
    if (ks == 32) { if (pa.old32) RICADI_RECT32(32, true); else RICADI_RECT32(32, false); }
    else { if (pa.old32) RICADI_RECT32(64, true); else RICADI_RECT32(64, false); }
#undef RICADI_RECT32
    return;
  }
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
          if (!(pa.out32 && pa.only32)) out[(size_t)row * ldo + col] = v;
          if (pa.out32) pa.out32[(size_t)grp * pa.gs32 + (size_t)row * ldo + col] = (float)v;
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
  if (pa.bmeta && bs == 32 && m == 16 && ldo == 16 && !pa.aggof && (s2.kstride == 32 || s2.kstride == 64) &&
      std::max(std::max(gso, s1.gs), s2.gs) * 8 < ((size_t)1 << 32)) {
#define RICADI_TWO32(K, H)                                                                                  \
  hipLaunchKernelGGL((block_two32_kernel<K, T, H>), grid, block, 0, st, gt, nblocks, pa.bmeta, pa.bm_stride, \
                     pa.bm_in, m1, s1, m2, s2, out, gso, pa)
    if (s2.kstride == 32) { if (s1.in16) RICADI_TWO32(32, true); else RICADI_TWO32(32, false); }
    else { if (s1.in16) RICADI_TWO32(64, true); else RICADI_TWO32(64, false); }
#undef RICADI_TWO32
    return;
  }
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
// The same product on the FP32 matrix cores (v_mfma_f32_16x16x4_f32: twice the rate of the FP64 form, no
// conversion of the FP32-stored inverse): the launch above is bound twice -- 0.53 of HBM on the inverses and 0.41 of
// the FP64 matrix peak on their FP64 products.  The coarse residual is rounded to FP32 on load, a wave accumulates its
// K slice (k / 8 terms) in FP32, the eight slices are summed in FP64.  Mirrored on scipy first (tools/schur_lab.py
// `sa+c32h`: iteration counts identical to the FP64 product at N = 30 / 58, NSE and DRE operators) -- it is the
// arithmetic of a flexible preconditioner whose inverse is FP32-stored already.  WRITTEN IN THE LAST SESSION OF ROUND 4
// WITHOUT GPU-MINUTES LEFT: never run on the device, off unless RICADI_COARSE32=1; DESIGN.md section 10a.
// Operand maps (cdna_hip_programming.md section 3): A[l & 15][k = l >> 4], B[k = l >> 4][l & 15] as in the FP64 form;
// C/D col = l & 15, row = 4 (l >> 4) + reg (the FP64 form: row = (l >> 4) + 4 reg).
__global__ __launch_bounds__(512) void dense_apply_tiled_f32mfma_kernel(GroupTab gt, int k, int m, GroupPtrsF Einvs,
                                                                        const double* __restrict__ rc,
                                                                        double* __restrict__ ec) {
  __shared__ float red[8][16][17];
  const int grp = gt.gid[blockIdx.z];
  const float* __restrict__ Einv = Einvs.p[grp];
  rc += (size_t)grp * k * m;
  ec += (size_t)grp * k * m;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int it = blockIdx.x, c0 = blockIdx.y * 16;
  const int kp = (k + 15) / 16;
  const int per = (kp + 7) / 8;
  const int ch0 = w * per, ch1 = min(kp, ch0 + per);
  const int col = c0 + r;
  f4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};   // one per chunk of a pass: no dependent MFMA pairs
  const size_t lane_off = (size_t)r * 16 + 4 * q;
  for (int ch = ch0; ch < ch1; ch += 2) {
    const bool two = ch + 1 < ch1;               // wave-uniform
    // all ten loads of a pass are requested before anything waits: the rc loads are unconditional on clamped
    // addresses (k >= 1, m >= 1) and zeroed by selects afterwards -- behind per-lane branches hipcc put a
    // vmcnt(0) after every one of them
    double d0[4], d1[4];
    const int colc = min(col, m - 1);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j0 = ch * 16 + 4 * q + t, j1 = j0 + 16;
      d0[t] = rc[(size_t)min(j0, k - 1) * m + colc];
      d1[t] = rc[(size_t)min(j1, k - 1) * m + colc];
    }
    const float* __restrict__ trow = Einv + ((size_t)it * kp + ch) * 256 + lane_off;
    const float4 a0 = *reinterpret_cast<const float4*>(trow);
    float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (two) a1 = *reinterpret_cast<const float4*>(trow + 256);
    float b0[4], b1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j0 = ch * 16 + 4 * q + t, j1 = j0 + 16;
      b0[t] = (j0 < k && col < m) ? (float)d0[t] : 0.f;
      b1[t] = (two && j1 < k && col < m) ? (float)d1[t] : 0.f;
    }
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1[0], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0[1], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1[2], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0[3], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1[3], acc1, 0, 0, 0);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[w][4 * q + e][r] = acc0[e] + acc1[e];
  __syncthreads();
  if (threadIdx.x < 256) {
    const int rr = threadIdx.x >> 4, cc = threadIdx.x & 15;
    double sum = 0.0;
#pragma unroll
    for (int t = 0; t < 8; ++t) sum += (double)red[t][rr][cc];
    const int row = it * 16 + rr;
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
                          int ldf, const double* rc, double* ec, bool f32_matrix_cores) {
  (void)ldf;   // tile-major storage (launch_to_f32_tiled)
  if (f32_matrix_cores) {
    if (k <= 0 || gt.ng <= 0) return;
    dim3 grid((k + 15) / 16, (m + 15) / 16, gt.ng);
    hipLaunchKernelGGL(dense_apply_tiled_f32mfma_kernel, grid, dim3(512), 0, st, gt, k, m, Einv, rc, ec);
    return;
  }
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
// K2p: the pressure step of the SIMPLE cycle in ONE launch (16-column panels, 32 x 32 Schur blocks):
//     t   = J z_v + (S Y)_p e - r_p          (CSR rows of J over the panel z, of the prolongated operator over
//                                              the coarse correction e; r_p from the FP64 or the FP16-stored vector)
//     z_p = Shat_b^-1 t[rows_b]               (FP64 MFMA 16x16x4 on the FP32- / FP64-stored block inverse)
// with the epilogue of the Schur sweep it replaces (plain copy for the J^T product of the last velocity sweep,
// coarse prolongation, FP32 copy).  Round 2 issued three dependent launches here (pressure rows of r - (S Y) e,
// J product, Schur sweep: 10 + 25 + 9 us at 16 groups, ~20 us of latency floor at one group).
// Round 4: the launch is a chain of dependent round trips per workgroup, not bytes (cfg2: 1 744 workgroups, all
// resident at once, 41 us; the 151 KB of z rows a block gathers come out of L2), so the chain is what was cut:
//   * 512 threads -- a 16-lane row per pressure row of the block, all 32 rows at once (two passes of 16 before);
//   * per (block, row) ONE 5-word record {row, J row range, (S Y) row range} at a fixed stride (ps_meta, built at
//     set_operator): block list -> row index -> row pointers were three dependent loads;
//   * the first (index, value) chunks of the J row and of the (S Y) row and r_p are requested together, the next
//     J chunk while the 16 gathers of the current one are in flight.
// Indices are loaded unconditionally (clamped to the row's last entry, the value zeroed instead): no exec-masked
// load blocks, cf. the scheduling rule in DESIGN.md.
// ---------------------------------------------------------------------------
template <class T, class RT, class ZT>
__global__ __launch_bounds__(512) void pressure_step_kernel(
    GroupTab gt, const int* __restrict__ meta, GroupPtrsT<T> invs,
    const int* __restrict__ jci, const double* __restrict__ jv, const ZT* __restrict__ z, size_t gsz,
    int with_sy, const int* __restrict__ syci, GroupPtrs syv, const double* __restrict__ ec, size_t gse,
    const RT* __restrict__ rp_, size_t gsr, double* __restrict__ out, size_t gso, ProlongArgs pa) {
  __shared__ double tl[32][17];
  const int grp = gt.gid[blockIdx.z];
  const T* __restrict__ inv = invs.p[grp] + (size_t)blockIdx.x * 1024;
  z += (size_t)grp * gsz;
  rp_ += (size_t)grp * gsr;
  out += (size_t)grp * gso;
  const double* __restrict__ sval = with_sy ? syv.p[grp] : nullptr;
  const double* __restrict__ ecg = with_sy ? ec + (size_t)grp * gse : nullptr;
  const int g = threadIdx.x & 15, il = threadIdx.x >> 4;        // column, row of the block
  const int* __restrict__ mt = meta + ((size_t)blockIdx.x * 32 + il) * 5;
  const int prow_ = mt[0], k0 = mt[1], k1 = mt[2], s0 = mt[3], s1 = mt[4];
  const bool live = prow_ >= 0;
  const int prow = live ? prow_ : 0;
  // round 1 of loads, all independent: r_p, first J chunk, first (S Y) chunk
  const RT rraw = rp_[(size_t)prow * 16 + g];
  const int klast = max(k1 - 1, k0);
  int nch = (k1 - k0 + 15) >> 4;
  nch = max(nch, __shfl_xor(nch, 16, 64));
  nch = max(nch, __shfl_xor(nch, 32, 64));
  int kk = k0 + g;
  int cn = jci[min(kk, klast)];
  double vn = jv[min(kk, klast)];
  bool okn = kk < k1;
  int sc = 0;
  double sv = 0.0;
  int nsch = 0;
  const int slast = max(s1 - 1, s0);
  if (with_sy) {
    nsch = (s1 - s0 + 7) >> 3;
    nsch = max(nsch, __shfl_xor(nsch, 16, 64));
    nsch = max(nsch, __shfl_xor(nsch, 32, 64));
    const int sk = s0 + (g & 7);
    sc = syci[min(sk, slast)];
    sv = syv.p[grp][min(sk, slast)];
    if (!(g < 8 && sk < s1)) sv = 0.0;
  }
  double acc = 0.0;
  for (int ch = 0; ch < nch; ++ch) {
    const int myc = cn;
    const double myv = okn ? vn : 0.0;
    if (ch + 1 < nch) {          // uniform per wave: the next chunk's indices travel while this one's gathers do
      kk += 16;
      cn = jci[min(kk, klast)];
      vn = jv[min(kk, klast)];
      okn = kk < k1;
    }
#define RICADI_PS_STEP(TT)                                  \
  {                                                         \
    const int c0 = bc16i<TT>(myc);                          \
    const double v0 = bc16d<TT>(myv);                       \
    acc = fma(v0, (double)z[(size_t)c0 * 16 + g], acc);     \
  }
    RICADI_FOR16(RICADI_PS_STEP)
#undef RICADI_PS_STEP
  }
  for (int ch = 0; ch < nsch; ++ch) {
    if (ch > 0) {
      const int sk = s0 + ch * 8 + (g & 7);
      sc = syci[min(sk, slast)];
      sv = sval[min(sk, slast)];
      if (!(g < 8 && sk < s1)) sv = 0.0;
    }
#define RICADI_PS_STEP(TT)                                  \
  {                                                         \
    const int c0 = bc16i<TT>(sc);                           \
    const double v0 = bc16d<TT>(sv);                        \
    acc = fma(v0, ecg[(size_t)c0 * 16 + g], acc);           \
  }
    RICADI_PS_STEP(0) RICADI_PS_STEP(1) RICADI_PS_STEP(2) RICADI_PS_STEP(3)
    RICADI_PS_STEP(4) RICADI_PS_STEP(5) RICADI_PS_STEP(6) RICADI_PS_STEP(7)
#undef RICADI_PS_STEP
  }
  tl[il][g] = live ? acc - (double)rraw : 0.0;
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
    const int ol = 16 * t + q + 4 * e;
    const int row = meta[((size_t)blockIdx.x * 32 + ol) * 5];
    if (row >= 0) {
      double v = acc4[e];
      if (pa.out2) pa.out2[(size_t)grp * pa.gs2 + (size_t)row * 16 + r] = v;
      if (pec) v += pec[(size_t)pa.aggof[row] * 16 + r];
      if (!(pa.out32 && pa.only32)) out[(size_t)row * 16 + r] = v;
      if (pa.out32) pa.out32[(size_t)grp * pa.gs32 + (size_t)row * 16 + r] = (float)v;
    }
  }
}
template <class T>
static void pressure_step_impl(hipStream_t st, const GroupTab& gt, int nblocks, const int* meta,
                               const GroupPtrsT<T>& inv, const int* jci, const double* jv, const double* z,
                               size_t gsz, bool with_sy, const int* syci, const GroupPtrs& syv, const double* ec,
                               size_t gse, const double* rp_, const _Float16* rp16, size_t gsr, double* out, size_t gso,
                               const ProlongArgs& pa, const float* zv32, size_t gsz32) {
  if (nblocks <= 0 || gt.ng <= 0) return;
  dim3 grid(nblocks, 1, gt.ng), block(512);
#define RICADI_PS(RT, ZT, rp, zp, gz)                                                                              \
  hipLaunchKernelGGL((pressure_step_kernel<T, RT, ZT>), grid, block, 0, st, gt, meta, inv, jci, jv, zp, gz,        \
                     with_sy ? 1 : 0, syci, syv, ec, gse, rp, gsr, out, gso, pa)
  if (rp16 && zv32) RICADI_PS(_Float16, float, rp16, zv32, gsz32);
  else if (rp16) RICADI_PS(_Float16, double, rp16, z, gsz);
  else if (zv32) RICADI_PS(double, float, rp_, zv32, gsz32);
  else RICADI_PS(double, double, rp_, z, gsz);
#undef RICADI_PS
}
void launch_pressure_step_b(hipStream_t st, const GroupTab& gt, int nblocks, const int* meta,
                            const GroupPtrsF& inv, const int* jci, const double* jv, const double* z,
                            size_t gsz, bool with_sy, const int* syci, const GroupPtrs& syv, const double* ec, size_t gse,
                            const double* rp_, const _Float16* rp16, size_t gsr, double* out, size_t gso,
                            const ProlongArgs& pa, const float* zv32, size_t gsz32) {
  pressure_step_impl(st, gt, nblocks, meta, inv, jci, jv, z, gsz, with_sy, syci, syv, ec, gse, rp_, rp16, gsr, out, gso,
                     pa, zv32, gsz32);
}
void launch_pressure_step_b(hipStream_t st, const GroupTab& gt, int nblocks, const int* meta,
                            const GroupPtrs& inv, const int* jci, const double* jv, const double* z,
                            size_t gsz, bool with_sy, const int* syci, const GroupPtrs& syv, const double* ec, size_t gse,
                            const double* rp_, const _Float16* rp16, size_t gsr, double* out, size_t gso,
                            const ProlongArgs& pa, const float* zv32, size_t gsz32) {
  pressure_step_impl(st, gt, nblocks, meta, inv, jci, jv, z, gsz, with_sy, syci, syv, ec, gse, rp_, rp16, gsr, out, gso,
                     pa, zv32, gsz32);
}

// ---- BF16 copies of the per-shift blocks (round 4): the two velocity sweeps are bandwidth bound on exactly these
// operands since they take their indices from one record (0.7 of the HBM roofline), and a block-Jacobi smoother does
// not feel an 8-bit mantissa (scipy mirror, N = 30 / 58, NSE and DRE operators: iteration counts unchanged to the
// last digit with BF16- or FP16-rounded blocks).  BF16 rather than FP16: FP32's exponent range, no per-block scale.
__global__ void to_bf16_kernel(size_t n, const double* __restrict__ src, uint16_t* __restrict__ dst) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned u = __float_as_uint((float)src[i]);
    dst[i] = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  }
}
void launch_to_bf16(hipStream_t st, size_t n, const double* src, uint16_t* dst) {
  if (!n) return;
  const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(to_bf16_kernel, dim3(grid), dim3(256), 0, st, n, src, dst);
}
bool launch_block_two32_h(hipStream_t st, const GroupTab& gt, int nblocks, const GroupPtrsH& m1, const Seg2& s1,
                          const GroupPtrsH& m2, const Seg2& s2, double* out, size_t gso, const ProlongArgs& pa,
                          bool f32_matrix_cores) {
  if (nblocks <= 0 || gt.ng <= 0) return true;
  if (!(pa.bmeta && !pa.aggof && (s2.kstride == 32 || s2.kstride == 64) &&
        std::max(std::max(gso, s1.gs), s2.gs) * 8 < ((size_t)1 << 32)))
    return false;
  dim3 grid((nblocks + 3) / 4, 1, gt.ng), block(256);
  if (f32_matrix_cores) {          // experimental (RICADI_SWEEP32=1), see block_two32_f32mfma_kernel
#define RICADI_TWO32F(K, H)                                                                                       \
  hipLaunchKernelGGL((block_two32_f32mfma_kernel<K, H>), grid, block, 0, st, gt, nblocks, pa.bmeta, pa.bm_stride, \
                     pa.bm_in, m1, s1, m2, s2, out, gso, pa)
    if (s2.kstride == 32) { if (s1.in16) RICADI_TWO32F(32, true); else RICADI_TWO32F(32, false); }
    else { if (s1.in16) RICADI_TWO32F(64, true); else RICADI_TWO32F(64, false); }
#undef RICADI_TWO32F
    return true;
  }
#define RICADI_TWO32(K, H)                                                                                         \
  hipLaunchKernelGGL((block_two32_kernel<K, uint16_t, H>), grid, block, 0, st, gt, nblocks, pa.bmeta, pa.bm_stride, \
                     pa.bm_in, m1, s1, m2, s2, out, gso, pa)
  if (s2.kstride == 32) { if (s1.in16) RICADI_TWO32(32, true); else RICADI_TWO32(32, false); }
  else { if (s1.in16) RICADI_TWO32(64, true); else RICADI_TWO32(64, false); }
#undef RICADI_TWO32
  return true;
}
bool launch_block_rect32_h(hipStream_t st, const GroupTab& gt, int ks, int nblocks, const GroupPtrsH& mats,
                           const double* in, size_t gsi, double* out, size_t gso, int subtract, const ProlongArgs& pa) {
  if (nblocks <= 0 || gt.ng <= 0) return true;
  if (!(pa.bmeta && (ks == 32 || ks == 64) && std::max(gsi, gso) * 8 < ((size_t)1 << 32) && !(pa.old32 && !pa.out32)))
    return false;
  const int nwaves = nblocks + (pa.aggof ? (pa.nextra + 31) / 32 : 0);
  dim3 grid((nwaves + 3) / 4, 1, gt.ng), block(256);
#define RICADI_RECT32(K, O)                                                                                         \
  hipLaunchKernelGGL((block_rect32_kernel<K, uint16_t, O>), grid, block, 0, st, gt, nblocks, pa.bmeta, pa.bm_stride, \
                     pa.bm_in, mats, in, gsi, out, gso, subtract, pa)
  if (ks == 32) { if (pa.old32) RICADI_RECT32(32, true); else RICADI_RECT32(32, false); }
  else { if (pa.old32) RICADI_RECT32(64, true); else RICADI_RECT32(64, false); }
#undef RICADI_RECT32
  return true;
}
void launch_pressure_step_h(hipStream_t st, const GroupTab& gt, int nblocks, const int* meta,
                            const GroupPtrsH& inv, const int* jci, const double* jv, bool with_sy, const int* syci,
                            const GroupPtrs& syv, const double* ec, size_t gse, const double* rp_, const _Float16* rp16,
                            size_t gsr, double* out, size_t gso, const ProlongArgs& pa, const float* zv32, size_t gsz32) {
  pressure_step_impl(st, gt, nblocks, meta, inv, jci, jv, (const double*)nullptr, 0, with_sy, syci, syv, ec, gse, rp_,
                     rp16, gsr, out, gso, pa, zv32, gsz32);
}

}  // namespace ricadi
