// GPU probe (not part of the product): what the all-to-all gradient hand-off among W workgroups costs under two
// protocols, with the slab size of BASELINE config 2 (9 f32x4 tiles per thread + a 256-float row + a tail = 38 KB):
//
//   A  (the product's, rounds 1-3)  stores -> s_waitcnt (acknowledged by the L2) -> barrier -> arrival counter
//      (atomic add, poll) -> barrier -> L1 invalidate -> every replica loads the W slabs in rank order
//   B  (candidate) the data carries its own readiness: triple-buffered slabs whose words the OWNER resets to an
//      all-ones word (a NaN no arithmetic produces) once everybody has read them, stores are fire-and-forget, a
//      consumer loads slab after slab and retries while any of ITS words still is the sentinel.  No counter, no
//      barrier, no acknowledged store on the critical path; the own slab comes from registers.
//
// Each iteration: spin `work` clocks (+ a pseudo-random skew per rank and iteration), hand-off, check the sums.
// Placement: all workgroups on XCC 0 (blocks 0, 8, 16, ...; ordinary memory, the XCC's L2 is the meeting point) or
// one per XCC round-robin (uncached memory).
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/handoff_probe.hip -o gpurun_out/handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32;
constexpr int NT = 9;
constexpr int XS = NT * 1024 + 512;
constexpr u32 SENT = 0xFFFFFFFFu;

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); }

__device__ __forceinline__ float gen(int it, int r, int q, int tid, int c) {
  return (float)((it * 7 + r * 3 + q * 5 + tid + c) & 255);  // small integers: every sum below is exact
}

__device__ __forceinline__ u32 max3u(u32 a, u32 b, u32 c) {
  u32 d;
  asm("v_max3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

struct Args {
  float* slabs;      // [3 or 2][W][XS]
  int* sync;         // [0] arrivals, [1] error flag, [8 + r] XCC of rank r
  long long* cyc;    // [W][2]: cycles inside the hand-off, cycles of the whole loop
  int W, iters, work, skew_mask, one_xcc, own_from_regs;
};

template <int PROTO>
__global__ __launch_bounds__(256) void handoff_kernel(Args a) {
  int rk;
  if (a.one_xcc) {
    if (blockIdx.x & 7) return;
    rk = blockIdx.x >> 3;
  } else {
    rk = blockIdx.x;
  }
  const int tid = threadIdx.x, W = a.W;
  if (tid == 0) a.sync[8 + rk] = xcc_id();
  __shared__ int dead_s;
  if (tid == 0) dead_s = 0;
  __syncthreads();
  long long in_handoff = 0;
  int errors = 0;
  bool dead = false;
  const long long t_begin = clock64();
  for (int it = 0; it < a.iters; ++it) {
    // ---- the step's compute: spin
    {
      const long long t = clock64();
      const int extra = ((u32)(it * 2654435761u + rk * 40503u) >> 7) & a.skew_mask;
      while (clock64() - t < a.work + extra) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    f32x4 g[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) g[q][c] = gen(it, rk, q, tid, c);
    const float gbias = gen(it, rk, NT, tid, 0), gs = 1.f;
    const long long t0 = clock64();
    constexpr int NBUF = PROTO == 0 ? 2 : 3;
    float* base = a.slabs + (long)(it % NBUF) * W * XS;
    float* own = base + (long)rk * XS;
    {
      f32x4* o4 = reinterpret_cast<f32x4*>(own);
#pragma unroll
      for (int q = 0; q < NT; ++q) o4[q * 256 + tid] = g[q];
      own[NT * 1024 + tid] = gbias;
      if (tid == 0) own[NT * 1024 + 256 + 5] = gs;
    }
    f32x4 s[NT];
    float sb = 0.f;
#pragma unroll
    for (int q = 0; q < NT; ++q) s[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (PROTO == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      if (tid == 0) {
        const int target = W * (it + 1);
        int seen = __hip_atomic_fetch_add(a.sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        int spins = 0;
        while (seen < target && !dead) {
          __builtin_amdgcn_s_sleep(1);
          seen = __hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (++spins > (1 << 20)) {
            __hip_atomic_store(a.sync + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dead_s = 1;
            break;
          }
        }
      }
      __syncthreads();
      dead = dead_s != 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      constexpr int RU = 4;
      for (int r0 = 0; r0 < W; r0 += RU) {
        f32x4 t[RU][NT];
        float tb[RU], tg[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const float* __restrict__ xr = base + (long)min(r0 + u, W - 1) * XS;
          const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(xr);
#pragma unroll
          for (int q = 0; q < NT; ++q) t[u][q] = x4[q * 256 + tid];
          tb[u] = xr[NT * 1024 + tid];
          tg[u] = xr[NT * 1024 + 256 + 5];
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          if (r0 + u < W) {
#pragma unroll
            for (int q = 0; q < NT; ++q) s[q] = s[q] + t[u][q] * tg[u];
            sb += tb[u] * tg[u];
          }
        }
      }
    } else {
      // ---- B: slab after slab, in rank order; RU slabs' loads in flight; the own slab from registers
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // lines of three iterations ago may sit in the L1
      constexpr int RU = 4;
      for (int r0 = 0; r0 < W; r0 += RU) {
        f32x4 t[RU][NT];
        float tb[RU], tg[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const int r = min(r0 + u, W - 1);
          if (a.own_from_regs && r == rk) continue;  // workgroup-uniform
          const float* __restrict__ xr = base + (long)r * XS;
          const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(xr);
#pragma unroll
          for (int q = 0; q < NT; ++q) t[u][q] = x4[q * 256 + tid];
          tb[u] = xr[NT * 1024 + tid];
          tg[u] = xr[NT * 1024 + 256 + 5];
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const int r = r0 + u;
          if (r >= W) continue;
          if (a.own_from_regs && r == rk) {
#pragma unroll
            for (int q = 0; q < NT; ++q) s[q] = s[q] + g[q] * gs;
            sb += gbias * gs;
            continue;
          }
          int spins = 0;
          while (true) {
            u32 m = max3u(__float_as_uint(tb[u]), __float_as_uint(tg[u]), 0u);
#pragma unroll
            for (int q = 0; q < NT; ++q) {
              m = max3u(m, __float_as_uint(t[u][q][0]), __float_as_uint(t[u][q][1]));
              m = max3u(m, __float_as_uint(t[u][q][2]), __float_as_uint(t[u][q][3]));
            }
            if (__builtin_amdgcn_ballot_w64(m == SENT) == 0 || dead) break;
            if (++spins > (1 << 18)) {
              __hip_atomic_store(a.sync + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              dead = true;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const float* __restrict__ xr = base + (long)r * XS;
            const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(xr);
#pragma unroll
            for (int q = 0; q < NT; ++q) t[u][q] = __builtin_nontemporal_load(x4 + q * 256 + tid);
            tb[u] = __builtin_nontemporal_load(xr + NT * 1024 + tid);
            tg[u] = __builtin_nontemporal_load(xr + NT * 1024 + 256 + 5);
          }
#pragma unroll
          for (int q = 0; q < NT; ++q) s[q] = s[q] + t[u][q] * tg[u];
          sb += tb[u] * tg[u];
        }
      }
      // ---- the owner resets the buffer of the previous iteration: everybody has read it (their slabs of THIS
      // iteration were stored after they had)
      float* prev = a.slabs + ((long)((it + 2) % 3) * W + rk) * XS;
      f32x4* p4 = reinterpret_cast<f32x4*>(prev);
      const float sf = __uint_as_float(SENT);
#pragma unroll
      for (int q = 0; q < NT; ++q) p4[q * 256 + tid] = (f32x4){sf, sf, sf, sf};
      prev[NT * 1024 + tid] = sf;
      if (tid == 0) prev[NT * 1024 + 256 + 5] = sf;
    }
    const long long t1 = clock64();
    in_handoff += t1 - t0;
    // ---- check
    if (!dead) {
#pragma unroll
      for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float e = 0.f;
          for (int r = 0; r < W; ++r) e += gen(it, r, q, tid, c);
          errors += (e != s[q][c]);
        }
      float e = 0.f;
      for (int r = 0; r < W; ++r) e += gen(it, r, NT, tid, 0);
      errors += (e != sb);
    }
    if (PROTO == 1) __builtin_amdgcn_s_waitcnt(0);  // the reset is performed before the next publication
  }
  const long long t_end = clock64();
  if (errors) atomicAdd(a.sync + 2, errors);
  if (tid == 0) {
    a.cyc[2 * rk] = in_handoff;
    a.cyc[2 * rk + 1] = t_end - t_begin;
  }
}

int main(int argc, char** argv) {
  int iters = 2000, work = 20000;
  if (argc > 1) iters = atoi(argv[1]);
  if (argc > 2) work = atoi(argv[2]);
  int dev_clock_khz = 0;
  CK(hipDeviceGetAttribute(&dev_clock_khz, hipDeviceAttributeClockRate, 0));
  printf("{\"iters\": %d, \"work_clocks\": %d, \"slab_bytes\": %d, \"rows\": [\n", iters, work, XS * 4);
  bool first = true;
  for (int one_xcc = 1; one_xcc >= 0; --one_xcc) {
    for (int W : {2, 4, 8}) {
      for (int skew_mask : {0, 2047}) {
        for (int proto = 0; proto < 3; ++proto) {  // 0: A, 1: B all slabs loaded, 2: B own slab from registers
          const size_t floats = (size_t)3 * W * XS;
          float* slabs = nullptr;
          if (one_xcc) CK(hipMalloc(&slabs, floats * 4));
          else CK(hipExtMallocWithFlags((void**)&slabs, floats * 4, hipDeviceMallocUncached));
          CK(hipMemset(slabs, 0xFF, floats * 4));
          int* sync = nullptr;
          long long* cyc = nullptr;
          CK(hipMalloc(&sync, 64 * 4));
          CK(hipMemset(sync, 0, 64 * 4));
          CK(hipMalloc(&cyc, 2 * W * 8));
          Args a{slabs, sync, cyc, W, iters, work, skew_mask, one_xcc, proto == 2};
          hipEvent_t e0, e1;
          CK(hipEventCreate(&e0));
          CK(hipEventCreate(&e1));
          const int grid = one_xcc ? 8 * (W - 1) + 1 : W;
          CK(hipEventRecord(e0));
          if (proto == 0) hipLaunchKernelGGL(handoff_kernel<0>, dim3(grid), dim3(256), 0, 0, a);
          else hipLaunchKernelGGL(handoff_kernel<1>, dim3(grid), dim3(256), 0, 0, a);
          CK(hipEventRecord(e1));
          CK(hipDeviceSynchronize());
          float ms = 0.f;
          CK(hipEventElapsedTime(&ms, e0, e1));
          std::vector<int> hs(64);
          std::vector<long long> hc(2 * W);
          CK(hipMemcpy(hs.data(), sync, 64 * 4, hipMemcpyDeviceToHost));
          CK(hipMemcpy(hc.data(), cyc, 2 * W * 8, hipMemcpyDeviceToHost));
          double in_h = 0, tot = 0;
          for (int r = 0; r < W; ++r) { in_h += (double)hc[2 * r] / iters / W; tot += (double)hc[2 * r + 1] / iters / W; }
          int xmask = 0;
          for (int r = 0; r < W; ++r) xmask |= 1 << hs[8 + r];
          printf("%s {\"placement\": \"%s\", \"W\": %d, \"skew_mask\": %d, \"protocol\": \"%s\", \"us_per_iter\": %.3f, "
                 "\"clocks_per_iter\": %.0f, \"clocks_in_handoff\": %.0f, \"timeout\": %d, \"wrong_sums\": %d, "
                 "\"xcc_mask\": %d}",
                 first ? "" : ",\n", one_xcc ? "one XCC, ordinary memory" : "one per XCC, uncached memory", W, skew_mask,
                 proto == 0 ? "A counter" : (proto == 1 ? "B sentinel" : "B sentinel, own slab from registers"),
                 ms * 1e3 / iters, tot, in_h, hs[1], hs[2], xmask);
          first = false;
          fflush(stdout);
          CK(hipFree(slabs));
          CK(hipFree(sync));
          CK(hipFree(cyc));
        }
      }
    }
  }
  printf("\n], \"clock_khz\": %d}\n", dev_clock_khz);
  return 0;
}
