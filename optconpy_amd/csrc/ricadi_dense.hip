// ricadi_dense.hip -- K5 / K6 / K4s: dense tall-skinny work on the FP64 matrix cores (GEMMs, CholQR2 panels,
// Householder TSQR, pivoted Cholesky) and the Cauchy recombination of an ADI sweep.
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



}  // namespace ricadi
