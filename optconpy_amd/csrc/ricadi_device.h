// ricadi_device.h -- device helpers shared by the kernel files (not installed).
#pragma once
#include <hip/hip_runtime.h>

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


}  // namespace ricadi
