// GPU probe: can ONE wave overlap independent VALU / transcendental / LDS work with its own in-flight
// v_mfma_f32_16x16x4_f32, or does overlap need a second wave on the SIMD?  Prints cycles per MFMA for
//   (a) MFMAs only, (b) + K independent v_fma per MFMA, (c) + K transcendentals, 1 wave and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NT>
__global__ void probe(float* out, long long* cyc, int iters) {
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f32x4& acc = (k == 0) ? acc0 : (k == 1) ? acc1 : (k == 2) ? acc2 : acc3;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);
#pragma unroll
      for (int q = 0; q < NT; ++q) v[q & 7] = __builtin_amdgcn_exp2f(v[q & 7] * 0.001f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = clock64();
  float s = 0;
  for (int q = 0; q < 8; ++q) s += v[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3] + s;
  if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)cyc, (unsigned long long)(t1 - t0));  // slowest wave
}

template <int NV, int NT>
void run(const char* name, int threads) {
  float* out; long long* cyc; hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipLaunchKernelGGL((probe<NV, NT>), dim3(1), dim3(threads), 0, 0, out, cyc, iters);
  hipMemset(cyc, 0, 8);
  hipLaunchKernelGGL((probe<NV, NT>), dim3(1), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s %d wave/SIMD: %.1f ticks per MFMA per wave = %.1f ticks per MFMA of the SIMD\n", name, threads / 256, (double)c / (iters * 4.0), (double)c / (iters * 4.0) / (threads / 256));
  fflush(stdout);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int threads : {256, 512, 1024}) {
    run<0, 0>("MFMA only", threads);
    run<2, 0>("MFMA + 2 independent v_fma", threads);
    run<4, 0>("MFMA + 4 independent v_fma", threads);
    run<6, 0>("MFMA + 6 independent v_fma", threads);
    run<8, 0>("MFMA + 8 independent v_fma", threads);
    run<12, 0>("MFMA + 12 independent v_fma", threads);
    run<0, 1>("MFMA + 1 v_exp (+1 mul)", threads);
    run<0, 2>("MFMA + 2 v_exp (+2 mul)", threads);
    run<3, 1>("MFMA + 3 v_fma + 1 v_exp", threads);
  }
  return 0;
}
