// Shared device/host helpers for libomnisafe_amd (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/omnisafe_amd.h"

#define OSA_WAVE 64

#define OSA_CHECK_LAUNCH()                          \
  do {                                              \
    hipError_t e__ = hipGetLastError();             \
    if (e__ != hipSuccess) return OSA_EHIP;         \
  } while (0)

#define OSA_REQUIRE(cond)             \
  do {                                \
    if (!(cond)) return OSA_EINVAL;   \
  } while (0)

static inline hipStream_t osa_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// "done once PER DEVICE": hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the current device only, and a
// process may drive more than one (the launch of a > 64 KB LDS kernel on a second device would otherwise fail)
struct OsaPerDeviceOnce {
  unsigned long long done = 0;  // bit d = device d (one node: <= 64 devices)
  bool need() const {
    int d = 0;
    (void)hipGetDevice(&d);
    return !((done >> (d & 63)) & 1ull);
  }
  void set() {
    int d = 0;
    (void)hipGetDevice(&d);
    done |= 1ull << (d & 63);
  }
};

// ActionScale.step (omnisafe/envs/wrapper.py:510-514), one element; no contraction so that every kernel that applies
// it (osa_action_scale_kernel, the policy step's fused epilogue) produces the same bits
__device__ __forceinline__ float osa_action_scale1(float a, float lo, float hi, float min_a, float max_a) {
#pragma clang fp contract(off)
  return lo + (hi - lo) * (a - min_a) / (max_a - min_a);
}

// ---- wave / block reductions (deterministic order) ------------------------------------------------
__device__ __forceinline__ double osa_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}
__device__ __forceinline__ float osa_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ float osa_wave_allsum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float osa_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float osa_wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
  return v;
}

// Wave-wide float sum on the DPP network (VALU only, no LDS round trips): inclusive scan with
// row_shr:1,2,4,8 inside each row of 16 lanes, then row_bcast:15 / row_bcast:31 across rows (the
// sequence LLVM's AMDGPU atomic optimizer emits for wave64 on gfx9); lane 63 holds the total, which is
// returned wave-uniformly.
__device__ __forceinline__ float osa_wave_sum_dpp(float v) {
  int x = __float_as_int(v);
#define OSA_DPP_ADD(CTRL, ROWMASK)                                                            \
  x = __float_as_int(__int_as_float(x) +                                                      \
                     __int_as_float(__builtin_amdgcn_update_dpp(0, x, (CTRL), (ROWMASK), 0xf, false)))
  OSA_DPP_ADD(0x111, 0xf);  // row_shr:1
  OSA_DPP_ADD(0x112, 0xf);  // row_shr:2
  OSA_DPP_ADD(0x114, 0xf);  // row_shr:4
  OSA_DPP_ADD(0x118, 0xf);  // row_shr:8
  OSA_DPP_ADD(0x142, 0xa);  // row_bcast:15 -> rows 1 and 3
  OSA_DPP_ADD(0x143, 0xc);  // row_bcast:31 -> rows 2 and 3
#undef OSA_DPP_ADD
  return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}

// Block-wide sum of a double; result returned to every thread.  `red` = LDS scratch of >= 17 doubles.
template <int THREADS>
__device__ __forceinline__ double osa_block_sum(double v, double* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = osa_wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) s += red[w];
    red[16] = s;
  }
  __syncthreads();
  return red[16];
}
