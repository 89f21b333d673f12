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
