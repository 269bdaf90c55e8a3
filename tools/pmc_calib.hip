// pmc_calib.hip -- known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in the access
// shapes of the saddle SpMM (K1): MI355X_MICROARCH.md says FETCH_SIZE reports 1/2 of the bytes of a wide coalesced
// 16-B-per-lane stream and that other widths are uncalibrated.  Not part of the product library.
//   hipcc -O3 --offload-arch=gfx950 tools/pmc_calib.hip -o tools/pmc_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o p -- tools/pmc_calib
// Every kernel reads a 2-GiB table exactly once (far beyond the 256-MiB infinity cache) and writes 1/16 of that:
//   stream16 : 16 B per lane, consecutive                       (the guide's calibrated case)
//   stream8  :  8 B per lane, consecutive
//   gather8  : each 16-lane group reads ONE 128-B row (8 B per lane) at a permuted row index   (K1's FP64 x gathers)
//   gather4  : each 16-lane group reads ONE  64-B row (4 B per lane) at a permuted row index   (K1's FP32 x gathers)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                          \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                           \
      exit(1);                                                                          \
    }                                                                                   \
  } while (0)

__global__ void stream16(const double2* __restrict__ t, size_t n, double* __restrict__ out) {
  double acc = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = t[i];
    acc += v.x + v.y;
  }
  out[blockIdx.x * (size_t)blockDim.x + threadIdx.x] = acc;
}
__global__ void stream8(const double* __restrict__ t, size_t n, double* __restrict__ out) {
  double acc = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += t[i];
  out[blockIdx.x * (size_t)blockDim.x + threadIdx.x] = acc;
}
// rows of 16 elements; a 16-lane group takes row perm[r]; every row is taken exactly once
template <class T>
__global__ void gather_rows(const T* __restrict__ t, const int* __restrict__ perm, size_t nrows, double* __restrict__ out) {
  const int g = threadIdx.x & 15;
  double acc = 0.0;
  const size_t ngrp = (size_t)gridDim.x * blockDim.x / 16;
  for (size_t r = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) / 16; r < nrows; r += ngrp)
    acc += (double)t[(size_t)perm[r] * 16 + g];
  out[blockIdx.x * (size_t)blockDim.x + threadIdx.x] = acc;
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  void* tab;
  CHK(hipMalloc(&tab, bytes));
  CHK(hipMemset(tab, 0, bytes));
  const int grid = 256 * 8, block = 256;
  double* out;
  CHK(hipMalloc((void**)&out, sizeof(double) * grid * block));
  // permutation of the rows: a multiplicative shuffle (odd multiplier modulo a power of two) -- far-apart consecutive rows
  const size_t rows8 = bytes / 128, rows4 = bytes / 64;
  std::vector<int> p8(rows8), p4(rows4);
  for (size_t i = 0; i < rows8; ++i) p8[i] = (int)((i * 2654435761ull) & (rows8 - 1));
  for (size_t i = 0; i < rows4; ++i) p4[i] = (int)((i * 2654435761ull) & (rows4 - 1));
  int *d8, *d4;
  CHK(hipMalloc((void**)&d8, rows8 * 4));
  CHK(hipMalloc((void**)&d4, rows4 * 4));
  CHK(hipMemcpy(d8, p8.data(), rows8 * 4, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d4, p4.data(), rows4 * 4, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(stream16, dim3(grid), dim3(block), 0, 0, (const double2*)tab, bytes / 16, out);
    hipLaunchKernelGGL(stream8, dim3(grid), dim3(block), 0, 0, (const double*)tab, bytes / 8, out);
    hipLaunchKernelGGL(gather_rows<double>, dim3(grid), dim3(block), 0, 0, (const double*)tab, d8, rows8, out);
    hipLaunchKernelGGL(gather_rows<float>, dim3(grid), dim3(block), 0, 0, (const float*)tab, d4, rows4, out);
  }
  CHK(hipDeviceSynchronize());
  printf("table %zu bytes read once per launch; index arrays: gather8 %zu bytes, gather4 %zu bytes; written per launch %zu bytes\n",
         bytes, rows8 * 4, rows4 * 4, sizeof(double) * grid * block);
  return 0;
}
