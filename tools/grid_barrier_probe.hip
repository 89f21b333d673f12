// GPU probe (not part of the product): what a software grid barrier costs among W co-resident workgroups that sit on
// ALL XCCs (one arrival counter per barrier instance in uncached memory, agent-scope atomics, one polling thread per
// workgroup) -- the price of a persistent rollout kernel's one global dependency per vector step (the normaliser's
// batch statistics).  Prints microseconds per barrier for W = 8 .. 256.
// Build: hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o tools/bin/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void barrier_kernel(int* cnt, int iters, int work, int payload, float* buf, int* err) {
  const int W = gridDim.x;
  __shared__ int dead;
  if (threadIdx.x == 0) dead = 0;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    const long long t = clock64();
    while (clock64() - t < work) __builtin_amdgcn_s_sleep(2);
    // optional payload: every workgroup publishes `payload` floats and reads everybody's after the barrier
    for (int e = threadIdx.x; e < payload; e += 256) buf[(size_t)((it & 1) * W + blockIdx.x) * payload + e] = (float)(it + e);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0 && !dead) {
      const int target = W * (it + 1);
      int seen = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
      int spins = 0;
      while (seen < target) {
        __builtin_amdgcn_s_sleep(1);
        seen = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (++spins > (1 << 22)) { *err = 1; dead = 1; break; }
      }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (payload) {
      float acc = 0.f;
      for (int r = 0; r < W; ++r)
        for (int e = threadIdx.x; e < payload; e += 256) acc += buf[(size_t)((it & 1) * W + r) * payload + e];
      if (acc == -1.f) *err = 2;
    }
  }
}

int main() {
  const int iters = 2000, work = 4000;
  for (int payload : {0, 128}) {
    for (int W : {8, 32, 64, 128, 256}) {
      int* cnt; int* err; float* buf;
      CK(hipExtMallocWithFlags((void**)&cnt, 256, hipDeviceMallocUncached));
      CK(hipExtMallocWithFlags((void**)&buf, (size_t)2 * 256 * 128 * 4 + 64, hipDeviceMallocUncached));
      CK(hipMalloc(&err, 4));
      CK(hipMemset(cnt, 0, 256)); CK(hipMemset(err, 0, 4));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      void* args[] = {&cnt, (void*)&iters, (void*)&work, &payload, &buf, &err};
      CK(hipEventRecord(e0));
      CK(hipLaunchCooperativeKernel((void*)barrier_kernel, dim3(W), dim3(256), args, 0, 0));
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
      int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      int khz = 0; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
      const double us = ms * 1e3 / iters, work_us = work / (khz / 1e3);
      printf("{\"workgroups\": %d, \"payload_floats_per_workgroup\": %d, \"us_per_iteration\": %.2f, \"us_of_work\": %.2f, \"us_per_barrier\": %.2f, \"err\": %d}\n",
             W, payload, us, work_us, us - work_us, herr);
      CK(hipFree(cnt)); CK(hipFree(buf)); CK(hipFree(err));
    }
  }
  return 0;
}
