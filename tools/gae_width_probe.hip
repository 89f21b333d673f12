// GPU probe (not part of the product): what the ACCESS SHAPE of the time-split GAE scan can reach on HBM, without
// any arithmetic.  A [T][N] float problem (4 input arrays + a flag byte read, 5 output arrays written = 37 B per
// transition, as osa_gae_chain_scan_kernel) is walked the way that kernel walks it: a workgroup of 8 waves owns a
// strip of `64 x V` envs (V floats per lane: 256 / 512 / 1024 contiguous bytes per row and wave) and 128 steps of it
// (wave w the steps 16 w .. 16 w + 15, backwards), every value is loaded into registers first and stored afterwards.
// Prints GB/s for V = 1, 2, 4.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gae_width_probe.hip -o tools/bin/gae_width_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

template <int V, int TC>
__global__ __launch_bounds__(512) void strip_kernel(const float* __restrict__ a0, const float* __restrict__ a1,
                                                    const float* __restrict__ a2, const float* __restrict__ a3,
                                                    const unsigned char* __restrict__ fl, float* __restrict__ o0,
                                                    float* __restrict__ o1, float* __restrict__ o2,
                                                    float* __restrict__ o3, float* __restrict__ o4, int T, int N) {
  typedef float vec __attribute__((ext_vector_type(V)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strips = N / (64 * V);
  const int strip = blockIdx.x % strips, level = blockIdx.x / strips;
  const long env0 = (long)strip * 64 * V + (long)lane * V;
  const int t_hi = T - 1 - (level * 8 + wave) * TC;
  if (t_hi < 0) return;
  vec r[TC], c[TC], vr[TC], vc[TC];
  unsigned char e[TC];
#pragma unroll
  for (int u = 0; u < TC; ++u) {
    const long i = (long)max(t_hi - u, 0) * N + env0;
    r[u] = *reinterpret_cast<const vec*>(a0 + i);
    c[u] = *reinterpret_cast<const vec*>(a1 + i);
    vr[u] = *reinterpret_cast<const vec*>(a2 + i);
    vc[u] = *reinterpret_cast<const vec*>(a3 + i);
    e[u] = fl[i];
  }
#pragma unroll
  for (int u = 0; u < TC; ++u) {
    if (t_hi - u < 0) continue;
    const long i = (long)(t_hi - u) * N + env0;
    const float k = e[u] ? 2.f : 1.f;
    *reinterpret_cast<vec*>(o0 + i) = r[u] * k;
    *reinterpret_cast<vec*>(o1 + i) = c[u] + vr[u];
    *reinterpret_cast<vec*>(o2 + i) = vr[u];
    *reinterpret_cast<vec*>(o3 + i) = vc[u] - r[u];
    *reinterpret_cast<vec*>(o4 + i) = c[u] * k;
  }
}

template <int V, int TC>
static void run(int T, int N, float** in, unsigned char* fl, float** out) {
  const int strips = N / (64 * V), levels = (T + 8 * TC - 1) / (8 * TC);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep)
    hipLaunchKernelGGL((strip_kernel<V, TC>), dim3(strips * levels), dim3(512), 0, 0, in[0], in[1], in[2], in[3], fl,
                       out[0], out[1], out[2], out[3], out[4], T, N);
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int rep = 0; rep < reps; ++rep)
    hipLaunchKernelGGL((strip_kernel<V, TC>), dim3(strips * levels), dim3(512), 0, 0, in[0], in[1], in[2], in[3], fl,
                       out[0], out[1], out[2], out[3], out[4], T, N);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, bytes = 37.0 * T * N;
  printf("{\"T\": %d, \"N\": %d, \"floats_per_lane\": %d, \"steps_per_wave\": %d, \"bytes_per_row_and_wave\": %d, "
         "\"us\": %.1f, \"GBps\": %.0f}\n", T, N, V, TC, 256 * V, us, bytes / us / 1e3);
}

int main() {
  for (int shape = 0; shape < 2; ++shape) {
    const int T = shape == 0 ? 4096 : 256, N = shape == 0 ? 4096 : 65536;
    const size_t M = (size_t)T * N;
    float* in[4];
    float* out[5];
    unsigned char* fl;
    for (int k = 0; k < 4; ++k) { CK(hipMalloc(&in[k], M * 4)); CK(hipMemset(in[k], 0, M * 4)); }
    for (int k = 0; k < 5; ++k) CK(hipMalloc(&out[k], M * 4));
    CK(hipMalloc(&fl, M));
    CK(hipMemset(fl, 0, M));
    run<1, 16>(T, N, in, fl, out);
    run<2, 16>(T, N, in, fl, out);
    run<2, 8>(T, N, in, fl, out);
    run<4, 8>(T, N, in, fl, out);
    run<4, 4>(T, N, in, fl, out);
    for (int k = 0; k < 4; ++k) CK(hipFree(in[k]));
    for (int k = 0; k < 5; ++k) CK(hipFree(out[k]));
    CK(hipFree(fl));
  }
  return 0;
}
