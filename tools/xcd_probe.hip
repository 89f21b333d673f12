// GPU probe (not part of the product): (1) which XCC does workgroup i of a 1-D grid land on, (2) latency
// of a two-workgroup message ping-pong (64 KB payload per hop optional) for
//    A: agent-scope protocol (write-through sc1 accesses + memory-side atomics), any placement
//    B: same-XCC protocol (plain stores + s_waitcnt, L2 atomics, buffer_inv sc0, plain loads)
// Build: hipcc --offload-arch=gfx950 -O3 tools/xcd_probe.hip -o gpurun_out/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); }

__global__ void where_kernel(int* out) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

// value of *p as the L2 holds it (an RMW executes in the L2; an idempotent __hip_atomic_fetch_or(p, 0) is
// folded into a load by the compiler, which may hit a stale line of the CU's vector L1)
__device__ __forceinline__ int l2_read(int* p) {
  int v;
  const int z = 0;
  asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p), "v"(z) : "memory");
  return v;
}

// two active workgroups (linear ids idA, idB) bounce a counter `iters` times; payload floats per hop
template <int MODE>
__global__ void pingpong_kernel(int idA, int idB, int* flag, float* buf, int payload, int iters, long long* cycles,
                                int* xcc, float* sink) {
  extern __shared__ float lds[];
  const int me = (blockIdx.x == idA) ? 0 : (blockIdx.x == idB) ? 1 : -1;
  if (me < 0) return;
  const int tid = threadIdx.x;
  if (tid == 0) xcc[me] = xcc_id();
  float acc = 0.f;
  __shared__ int dead;
  if (tid == 0) dead = 0;
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (dead) break;
    const int turn = it & 1;  // who sends
    float* slab = buf + (size_t)(it & 1) * payload;
    if (turn == me) {
      for (int e = tid; e < payload; e += blockDim.x) {
        if (MODE == 0) __hip_atomic_store(slab + e, (float)(it + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else slab[e] = (float)(it + e);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        if (MODE == 0) __hip_atomic_store(flag, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_exchange(flag, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else {
      if (tid == 0) {
        int spins = 0;
        while (true) {
          int v;
          if (MODE == 0) v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else v = l2_read(flag);
          if (v >= it + 1) break;
          if (++spins > (1 << 16)) { dead = 1; break; }
        }
      }
      __syncthreads();
      if (MODE == 1) asm volatile("buffer_inv sc0" ::: "memory");
      for (int e = tid; e < payload; e += blockDim.x) {
        float v;
        if (MODE == 0) v = __hip_atomic_load(slab + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else v = slab[e];
        if (v != (float)(it + e)) acc += 1.f;  // stale / wrong data counter
      }
    }
    __syncthreads();
  }
  long long t1 = clock64();
  acc = acc;  // number of mismatches seen by this thread
  atomicAdd(sink + me, acc + (dead ? 1e9f : 0.f));
  if (tid == 0) cycles[me] = t1 - t0;
}

int main() {
  int* d_out; hipMalloc(&d_out, 256 * sizeof(int));
  hipFuncSetAttribute((const void*)where_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipLaunchKernelGGL(where_kernel, dim3(64), dim3(256), 150 * 1024, 0, d_out);
  std::vector<int> h(64); hipMemcpy(h.data(), d_out, 64 * sizeof(int), hipMemcpyDeviceToHost);
  printf("xcc of workgroup i (150 KB LDS each):");
  for (int i = 0; i < 64; ++i) printf(" %d", h[i]);
  printf("\n"); fflush(stdout);
  int* flag; float* buf; long long* cyc; int* xcc; float* sink;
  hipMalloc(&flag, 4); hipMalloc(&buf, 2 * 65536 * 4); hipMalloc(&cyc, 16); hipMalloc(&xcc, 8); hipMalloc(&sink, 8);
  hipFuncSetAttribute((const void*)pingpong_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipFuncSetAttribute((const void*)pingpong_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  const int iters = 500;
  for (int payload : {0, 256, 9472}) {
    for (int mode = 0; mode < 2; ++mode) {
      for (int idB : {8, 1}) {  // 8: same XCC as workgroup 0 under round-robin dispatch; 1: a different XCC
        hipMemset(flag, 0, 4); hipMemset(sink, 0, 8); hipMemset(buf, 0, 2 * 65536 * 4);
        if (mode == 0) hipLaunchKernelGGL(pingpong_kernel<0>, dim3(16), dim3(256), 150 * 1024, 0, 0, idB, flag, buf, payload, iters, cyc, xcc, sink);
        else hipLaunchKernelGGL(pingpong_kernel<1>, dim3(16), dim3(256), 150 * 1024, 0, 0, idB, flag, buf, payload, iters, cyc, xcc, sink);
        hipDeviceSynchronize();
        long long c[2]; int x[2]; float s[2];
        hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost); hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost); hipMemcpy(s, sink, 8, hipMemcpyDeviceToHost);
        printf("payload %5d floats, protocol %s, workgroups 0 and %d (xcc %d, %d): %.0f clock64 ticks per hop, mismatches %.0f\n",
               payload, mode == 0 ? "A(agent/sc1)" : "B(same-XCC/L2)", idB, x[0], x[1], (double)c[0] / iters, s[0] + s[1]); fflush(stdout);
      }
    }
  }
  return 0;
}
