// Persistent PPO-Lag update pass for WIDE observations, first layer SPLIT OVER COOPERATING CUs
// (BASELINE config 4: SafetyHumanoidVelocity, 376 / 17).  Same contract as osa_ppo_wide_pass
// (wide_pass_kernel.hip): ONE launch = one whole pass of PolicyGradient._update's inner loop
// (policy_gradient.py:366-382), `nmb` dependent minibatch optimiser steps of <= 64 rows, three networks.
//
// osa_wide_pass_kernel runs a network on ONE CU and is bound by what that CU can move per step: W1 (96 KB at
// 376 inputs), its two Adam moments and the gradient of the step in flight stream through one vector L1 --
// 45 k of the 89 k cycles of a step, with 14 us of MFMA issue in a 37 us step.  Here a network is C + 1
// workgroups on C + 1 CUs (cooperative launch):
//   * helper c (C = ceil(KB / 6) of them) OWNS a slice of <= 6 K blocks (<= 96 input features) of W1 for the
//     whole pass: the slice in LDS (25 KB), its Adam moments in registers (24 elements per lane) -- the
//     layer-1 half of osa_ppo_pass_kernel.  Per step it gathers its 96 columns of the minibatch rows,
//     computes the PARTIAL pre-activation W1[:, slice] x[slice] (96 MFMAs per wave) and publishes it; after
//     the leader's dz1 arrives: dW1[:, slice] = dz1 x[slice]^T (96 MFMAs per wave, operands swapped so that a
//     lane's accumulators are exactly the elements it owns), + 2 c w (critics), its share of |g|^2;
//   * the leader owns W2, W3, the biases and log_std (LDS master copy, as the one-CU kernel): sums the C
//     partial pre-activations in slice order, runs the rest of the forward pass, the loss and the backward
//     pass, publishes dz1, computes dW2 / dW3 / bias gradients and its share of |g|^2;
//   * the C + 1 squared-norm shares meet in an all-gather (every workgroup sums the slots in the same order:
//     one clip factor, bit-identical everywhere); then everybody applies Adam to what it owns.
// Three hand-offs per step (partials -> leader, dz1 -> helpers, norm shares -> all) through the exchange buffer.
// Placement (OsaSplitArgs::local): workgroup b of a grid runs on XCC b mod 8, so with local = 1 the launch is
// 8 (C + 1) blocks of which block net + 8 role works: a network's workgroups share ONE XCC and the buffer is
// ordinary memory served by that XCC's L2; with local = 0 consecutive blocks (all XCCs) and an UNCACHED buffer
// (osa_dp_exchange_alloc: stores are performed at the device-coherent level).  Either way a hand-off is "my stores
// are performed (s_waitcnt vmcnt(0)), LDS barrier, relaxed agent-scope flag store" on one side and on the other
// "poll, LDS barrier, L1 invalidate (buffer_inv sc1: the vector L1 keeps lines even of uncached memory), loads"
// (leader) or "every lane polls the flag and requests its own fragments right behind it with cache-bypassing
// loads: one round trip" (helpers); the squared-norm shares travel with their step counter in one 8-byte word.
// No L2 write-back anywhere.  Flags are step counters (monotonic within a launch); the one-XCC placement is
// verified before anything is modified (all workgroups return untouched otherwise) and a peer that never
// arrives raises a sticky flag instead of hanging the device.
//
// Arithmetic: the fragment algebra and loss code of osa_wide_pass_kernel; the layer-1 pre-activation is
// the sum of C partial sums instead of one chain over K (float32 re-association, ~1e-7 relative), and the
// critics' L2 term is added element-wise (the helpers hold their weights) instead of through the algebraic
// identity.  Parity: tests/test_mlp_gpu.py (vs the per-step kernels), tests/test_config_shapes_gpu.py::config4.
#include "mlp_device.h"

#define SSLD 68    // leading dimension (floats) of every [row][64 + pad] LDS tile
#define SXLD 100   // leading dimension of the helper's W1 slice [64][96 + pad]
#define SNSTAT 16
#define SCMAX 6    // helpers per network (6 x 96 = 576 >= 512 inputs)
#define SKQ 6      // K blocks per helper
// exchange buffer (floats): [128] flag words, then per network SXNET floats
#define SX_PART 0                          // [SCMAX][4096] partial pre-activations (fragment order)
#define SX_DZ (SCMAX * 4096)               // [2 step parity][64][64] dz1, row = feature; words not yet published hold SX_SENT
#define SX_NORM (SX_DZ + 2 * 4096)         // [SCMAX + 1][2] 8-byte words {float share, int step}: |g|^2, |w|^2
#define SX_SENT 0xFFFFFFFFu                // "not published": a NaN no arithmetic produces (host memset 0xFF per launch)
#define SXNET (SX_NORM + 32)
#define SF_PART 0                          // flag words of a network: [32 net + ..]
#define SF_DZ 8
#define SF_NORM 16
#define SF_XCC 25                          // bit mask of the XCCs the network's workgroups ran on
#define SF_ERR 96                          // sticky: a peer never arrived (1) / a network on two XCCs (2)
#define SF_ARRIVE 97                       // arrivals of the placement check (zeroed per launch)

struct OsaSplitHp {
  float clip, entropy_coef, critic_norm_coef, max_grad_norm;
  float lr_actor, lr_critic, beta1, beta2, adam_eps;
  int use_critic_norm, use_max_grad_norm, use_cost;
};

struct OsaSplitArgs {
  OsaNet nd;
  float* params;   // [3][P] padded global layout
  float* adam_m;   // [3][P]
  float* adam_v;   // [3][P]
  int* adam_step;  // [3]
  const float* obs;
  int ld_obs;
  const float* act;
  int ld_act;
  const float* logp;
  const float* tgt_r;
  const float* tgt_c;
  const float* adv_r;
  const float* adv_c;
  const long* perm;  // [M] sample rows of the whole pass (nullptr = identity)
  long M;
  int B;    // minibatch size (<= 64); the last minibatch may be smaller
  int nmb;  // minibatches in this launch
  const float* lagrange;
  OsaSplitHp hp;
  int loss_kind;
  int nets_mask;
  float* stats;  // [nmb][SNSTAT]
  float* xch;    // uncached exchange buffer (osa_ppo_split_pass_xch_floats)
  int C;         // helpers per network
  int local;     // 1: one XCC per network, cached exchange buffer (see the kernel)
  // data-parallel form (DP instantiation, osa_ppo_split_dp_pass): `world` virtual ranks, each a full set of
  // 3 (C + 1) workgroups working on ITS rows (rank r: rows r M .. r M + M - 1 of the all-gathered arrays,
  // permutation perm[r M ..]) with its own intra-rank exchange region (xch + r rank_xch); after a rank's clip
  // factor is known, the `world` owners of the same parameters (helper c of every rank / the leaders) average
  // their locally clipped gradients through dpx (clip-then-average, policy_gradient.py:437-442 +
  // distributed.py:167-198) and every replica applies the same Adam step: nothing but gradients crosses between
  // the replicas and they stay bit-identical.
  int world;
  long rank_xch;  // floats between the intra-rank regions of consecutive ranks
  // cross-rank exchange: [SDPH] header floats (unused here), then [2 parity][3][SCMAX + 1][world][SDPW] slabs
  float* dpx;
  // its header: [SDPH] ints (arrival counters [8 net + role], XCC masks [32 + group], placement arrivals [96]) --
  // ALWAYS in the uncached region, also when the slabs are ordinary memory (dp_place 1): words that are reset by a
  // memset per launch and then counted up by workgroups on all XCCs have no business in a write-back cache
  int* dp_hdr;
  // 0: rank-major blocks spread over the XCCs, dpx uncached like xch (every replica then reads the `world` slabs of
  //    its owners from the device-coherent level: world^2 x 25 KB per owner group and step -- 24 MB per step at
  //    376 / 17 and world 8, which is what bounds that variant);
  // 1: the `world` owners of the same parameters (one "group" = (network, role)) are placed on ONE XCC (block
  //    b -> XCC b mod 8: group g on XCC g mod 8) and dpx is ordinary memory served by that XCC's L2; the
  //    intra-rank hand-offs (partials, dz1, norm shares) cross XCCs through the uncached xch.  Placement verified
  //    before anything is modified (sticky flag 2 otherwise: repeat with 0).
  int dp_place;
};

#define SDPW (SKQ * 1024 + 256 + 16)  // one owner's gradient share: <= 6 tiles x 256 x 4, bias-likes, tail
#define SDPH 256                      // header words of the cross-rank exchange buffer
#ifndef OSA_SPLIT_DP_RU
#define OSA_SPLIT_DP_RU 2  // (4: 129 spilled VGPRs in the 17-action instantiation; 2: see DESIGN.md)
#endif

#ifdef OSA_SPLIT_CLOCKS
#define STICK(k)                                   \
  do {                                             \
    if (tid == 0) {                                \
      const long long now_ = clock64();            \
      sdbg[k] += now_ - slast;                     \
      slast = now_;                                \
    }                                              \
  } while (0)
#else
#define STICK(k) do { } while (0)
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope release, which on
// gfx950 waits for EVERY outstanding vector-memory operation (one counter for loads and stores): it would
// drain the next step's gathers, issued early on purpose, at each barrier of the step.
__device__ __forceinline__ void osa_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// my stores to the exchange buffer have been performed (uncached memory: at the device-coherent level)
__device__ __forceinline__ void osa_xch_release() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
}

// after a wait: drop what this CU's vector L1 still holds of the exchange buffer (it keeps lines of uncached
// memory between two reads of the same address when little else is loaded in between: stale partials / dz1
// were observed with small rollouts); an L1 invalidate, no L2 write-back
__device__ __forceinline__ void osa_xch_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }

__device__ __forceinline__ unsigned osa_max3u(unsigned a, unsigned b, unsigned c) {
  unsigned d;
  asm("v_max3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// 16-byte load that bypasses the vector L1 and is served at the coherence point of the exchange buffer (sc0 sc1);
// the caller waits for it (s_waitcnt vmcnt) before it uses the value
__device__ __forceinline__ f32x4 osa_load_bypass4(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// lane `tid < n` waits until flag[tid] >= target (bounded: raises the sticky flag and gives up for good)
__device__ __forceinline__ void osa_xch_wait(int* flags, int n, int target, int tid, int* err, bool& dead) {
  if (tid < n && !dead) {
    int spins = 0;
    while (__hip_atomic_load(flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 20)) {
        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dead = true;
        break;
      }
    }
  }
}

// Squared-norm shares travel WITH their step counter in one 8-byte word (a relaxed agent-scope atomic): the
// reader polls the data itself -- no flag -> invalidate -> data round trip in the all-gather.
__device__ __forceinline__ void osa_slot_put(unsigned long long* slot, float v, int cnt) {
  const unsigned long long w = ((unsigned long long)(unsigned)cnt << 32) | (unsigned long long)__float_as_uint(v);
  __hip_atomic_store(slot, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float osa_slot_wait(unsigned long long* slot, int cnt, int* err, bool& dead) {
  unsigned long long w = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int spins = 0;
  while ((int)(w >> 32) < cnt && !dead) {
    __builtin_amdgcn_s_sleep(1);
    w = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (++spins > (1 << 20)) {
      __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      dead = true;
    }
  }
  return __uint_as_float((unsigned)w);
}

// Average of the `W` ranks' locally clipped gradients of ONE owner's parameters (data-parallel form): publish the
// own share (NT tiles of one f32x4 per thread, optionally one bias-like scalar per thread, then the clip factor
// and -- leaders -- the step's statistics in the tail), arrive at the owners' counter, wait for the W arrivals of
// this step, then sum g_r * clip_r in RANK ORDER (the same order on every replica: bit-identical results) / W.
// slabs: [W][SDPW] of this (step parity, network, role); uncached memory, so "published" = stores performed.
template <int NT>
__device__ __forceinline__ void osa_split_dp_average(f32x4 (&g)[NT], float& gb, float coef, const float* tail5,
                                                     float* slabs, int rk, int W, int* cnt, int target, int tid,
                                                     int* err, bool& dead, long long* clk = nullptr) {
#ifdef OSA_SPLIT_CLOCKS
  long long c0 = clock64();
#define DTICK(k) do { if (clk && tid == 0) { const long long n_ = clock64(); clk[k] += n_ - c0; c0 = n_; } } while (0)
#else
#define DTICK(k) do { } while (0)
#endif
  {
    float* own = slabs + (long)rk * SDPW;
    f32x4* o4 = reinterpret_cast<f32x4*>(own);
#pragma unroll
    for (int q = 0; q < NT; ++q) o4[q * 256 + tid] = g[q];
    own[SKQ * 1024 + tid] = gb;
    if (tid == 0) own[SKQ * 1024 + 256] = coef;
    if (tail5 != nullptr && tid >= 1 && tid <= 5) own[SKQ * 1024 + 256 + tid] = tail5[tid - 1];
  }
  osa_xch_release();
  osa_lds_barrier();  // every thread's stores are performed
  DTICK(0);
  if (tid == 0 && !dead) {
    int seen = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    int spins = 0;
    while (seen < target) {
      __builtin_amdgcn_s_sleep(1);
      seen = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (++spins > (1 << 20)) {  // a peer never arrived: sticky flag, never hang the device
        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dead = true;
        break;
      }
    }
  }
  osa_lds_barrier();
  osa_xch_acquire();
  DTICK(1);
  f32x4 s[NT];
  float sb = 0.f;
#pragma unroll
  for (int q = 0; q < NT; ++q) s[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int RU = OSA_SPLIT_DP_RU;  // ranks per trip: their loads are all in flight together
  for (int r0 = 0; r0 < W; r0 += RU) {
    f32x4 t[RU][NT];
    float tb[RU], tg[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const float* xr = slabs + (long)min(r0 + u, W - 1) * SDPW;
      const f32x4* x4 = reinterpret_cast<const f32x4*>(xr);
#pragma unroll
      for (int q = 0; q < NT; ++q) t[u][q] = x4[q * 256 + tid];
      tb[u] = xr[SKQ * 1024 + tid];
      tg[u] = xr[SKQ * 1024 + 256];
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      if (r0 + u < W) {  // workgroup-uniform
#pragma unroll
        for (int q = 0; q < NT; ++q) s[q] = s[q] + t[u][q] * tg[u];
        sb += tb[u] * tg[u];
      }
    }
  }
  const float invW = 1.f / (float)W;
#pragma unroll
  for (int q = 0; q < NT; ++q) g[q] = s[q] * invW;
  gb = sb * invW;
  DTICK(2);
#undef DTICK
}

template <int OT, bool DP>
__global__ __launch_bounds__(256, 1) void osa_wide_split_kernel(OsaSplitArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int H = 64, HT = 4, OUTP = 16 * OT;
  const OsaNet& nd = a.nd;
  const int C = a.C;
  // placement: workgroup b of a grid runs on XCC b mod 8 (tools/xcd_probe.hip).  `local`: the C + 1 workgroups of
  // a network are the blocks b = net + 8 role, i.e. they share ONE XCC and its L2, and the exchange buffer is
  // ordinary cached memory (a hand-off then costs L2 round trips, not trips to the device-coherent level);
  // otherwise consecutive blocks (spread over the XCCs) and an uncached buffer.  role 0: leader, 1 + c: helper c
  int net, role, rk = 0;
  if constexpr (DP) {
    if (a.dp_place == 1) {  // owner group g = (network, role) on XCC g mod 8: blocks g mod 8 + 8 ((g / 8) world + rank)
      const int slot = blockIdx.x >> 3, gid = (blockIdx.x & 7) + 8 * (slot / a.world);
      if (gid >= 3 * (C + 1)) return;
      rk = slot % a.world;
      net = gid / (C + 1);
      role = gid - net * (C + 1);
    } else {  // rank-major: the 3 (C + 1) workgroups of rank rk are consecutive blocks
      const int per = 3 * (C + 1), b = blockIdx.x % per;
      rk = blockIdx.x / per;
      net = b / (C + 1);
      role = b - net * (C + 1);
    }
  } else if (a.local == 1) {
    net = blockIdx.x & 7;
    role = blockIdx.x >> 3;
    if (net >= 3) return;
  } else {  // (local == 3: test hook -- the one-XCC protocol on the spread grid, so that the placement check trips)
    net = blockIdx.x / (C + 1);
    role = blockIdx.x - net * (C + 1);
  }
  if (!((a.nets_mask >> net) & 1)) return;
  // data-parallel form: this rank's rows, permutation and intra-rank exchange region
  const long roff = DP ? (long)rk * a.M : 0;
  const float* __restrict__ obs_p = a.obs + roff * a.ld_obs;
  const float* __restrict__ act_p = a.act + roff * a.ld_act;
  const float* __restrict__ logp_p = a.logp + roff;
  const float* __restrict__ advr_p = a.adv_r + roff;
  const float* __restrict__ advc_p = a.adv_c + roff;
  const long* __restrict__ perm_p = a.perm ? a.perm + roff : nullptr;
  float* const xch_r = a.xch + (DP ? (long)rk * a.rank_xch : 0);
  const int W = DP ? a.world : 1;
  // cross-rank exchange of this (network, role): arrival counter + [2 parity][world][SDPW] slabs
  int* dp_cnt = DP ? a.dp_hdr + 8 * net + role : nullptr;
  auto dp_slabs = [&](int mb) -> float* {
    return a.dpx + SDPH + ((((long)(mb & 1) * 3 + net) * (SCMAX + 1) + role) * W) * SDPW;
  };
  const int KB = nd.KB, INP = nd.INP, P = nd.P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int i = j, cc = j;
  float* __restrict__ gp = a.params + (long)net * P;
  float* __restrict__ gm = a.adam_m + (long)net * P;
  float* __restrict__ gv = a.adam_v + (long)net * P;
  int* flags = reinterpret_cast<int*>(xch_r) + 32 * net;
  int* err = reinterpret_cast<int*>(a.xch) + SF_ERR;  // (one sticky word for the launch: rank 0's region)
  float* xn = xch_r + 128 + (long)net * SXNET;
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(xn + SX_NORM);  // [2 (C + 1)] {value, step}
  const bool critic = net != 0;
  const bool l2 = critic && a.hp.use_critic_norm;
  const float c2 = 2.f * a.hp.critic_norm_coef;
  const float beta1 = a.hp.beta1, beta2 = a.hp.beta2, aeps = a.hp.adam_eps;
  bool dead = false;
  if (a.local) {
    // Verify the placement BEFORE anything is modified: every workgroup ORs its XCC bit into its network's
    // SF_XCC word and counts itself in at SF_ARRIVE; when all (active networks x (C + 1)) have arrived every
    // network's mask must hold ONE bit.  Otherwise EVERY workgroup returns with parameters, Adam state and step
    // counters untouched and the sticky word says why (2: a network on two XCCs, 1: somebody never arrived) --
    // the caller repeats the pass with local = 0 (update.py).
    int* s_why = reinterpret_cast<int*>(smem);  // (the dynamic LDS is not in use yet; a static __shared__ variable
    // on top of the 160 KB dynamic limit makes hipFuncSetAttribute refuse the kernel)
    if (tid == 0) {
      int* base = reinterpret_cast<int*>(a.xch);
      const int xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7;  // HW_REG_XCC_ID[3:0]
      __hip_atomic_fetch_or(flags + SF_XCC, 1 << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int expect = (C + 1) * __builtin_popcount(a.nets_mask & 7);
      int v = __hip_atomic_fetch_add(base + SF_ARRIVE, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) + 1;
      int spins = 0, why = 0;
      while (v < expect) {
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(base + SF_ARRIVE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (++spins > (1 << 21)) { why = 1; break; }
      }
      for (int n = 0; n < 3 && why == 0; ++n) {
        const int mask = __hip_atomic_load(base + 32 * n + SF_XCC, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if ((mask & (mask - 1)) != 0) why = 2;
      }
      if (why) __hip_atomic_store(err, why, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (a late arriver sees a full count although an earlier one has already given up: the sticky word decides)
      if (!why) why = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *s_why = why;
    }
    __syncthreads();
    const bool bad_placement = *s_why != 0;
    __syncthreads();
    if (bad_placement) return;
  }
  if constexpr (DP) {
    if (a.dp_place != 0) {  // (3: test hook -- verification on the rank-major grid, where it must trip)
      // every owner group must sit on ONE XCC (its exchange slabs are ordinary memory in that XCC's L2): verified
      // before anything is modified, as above; otherwise everybody returns untouched with the sticky word at 2
      int* s_why = reinterpret_cast<int*>(smem);
      if (tid == 0) {
        int* hdr = a.dp_hdr;
        const int G = 3 * (C + 1), gid = net * (C + 1) + role;
        const int xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7;  // HW_REG_XCC_ID[3:0]
        __hip_atomic_fetch_or(hdr + 32 + gid, 1 << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int expect = a.world * (C + 1) * __builtin_popcount(a.nets_mask & 7);
        int v = __hip_atomic_fetch_add(hdr + 96, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) + 1;
        int spins = 0, why = 0;
        while (v < expect) {
          __builtin_amdgcn_s_sleep(1);
          v = __hip_atomic_load(hdr + 96, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (++spins > (1 << 21)) { why = 1; break; }
        }
        for (int q = 0; q < G && why == 0; ++q) {
          const int mask = __hip_atomic_load(hdr + 32 + q, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          if ((mask & (mask - 1)) != 0) why = 2;
        }
        if (why) __hip_atomic_store(err, why, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!why) why = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_why = why;
      }
      __syncthreads();
      const bool bad_placement = *s_why != 0;
      __syncthreads();
      if (bad_placement) return;
    }
  }
#ifdef OSA_SPLIT_CLOCKS
  long long sdbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long dclk[3] = {0, 0, 0};  // data-parallel average: publish + release | arrive + wait + acquire | slab reads
  long long slast = clock64();
#endif

  if (role > 0) {
    // =====================================================================================================
    // helper: K blocks kb0 .. kb0 + nkb - 1 of W1
    // =====================================================================================================
    const int hc = role - 1;
    const int kb0 = hc * SKQ, nkb = min(SKQ, KB - kb0);  // the last slice may be short (zero blocks)
    float* sW = smem;                     // [H][SXLD]   W1 slice, element (feature f, local input k)
    float* sX = sW + H * SXLD;            // [96][SSLD]  X^T of the step: element (local input k, sample c)
    float* red = sX + 16 * SKQ * SSLD;    // [16]
    float* sBC = red + 16;                // [nmb][2] Adam's bias corrections of every step (as the leader's table)
    const int ld_obs = a.ld_obs, obs_dim = nd.obs_dim;
    auto load_chunk = [&](const float* __restrict__ xr, int col0) -> f32x4 {
      const int cl = (col0 + 4 <= ld_obs) ? col0 : ld_obs - 4;
      return *reinterpret_cast<const f32x4*>(xr + cl);
    };
    auto mask_chunk = [&](f32x4 v, int col0) -> f32x4 {
      v.x = (col0 + 0 < obs_dim) ? v.x : 0.f;
      v.y = (col0 + 1 < obs_dim) ? v.y : 0.f;
      v.z = (col0 + 2 < obs_dim) ? v.z : 0.f;
      v.w = (col0 + 3 < obs_dim) ? v.w : 0.f;
      return v;
    };
    // the W1 elements this lane owns (D layout of the transposed weight-gradient tiles): feature row
    // 16 wave + cc, local inputs 16 q + 4 g .. + 3 for q < SKQ
    const int own_lds = (16 * wave + cc) * SXLD + 4 * g;
    const int own_glb = nd.oW1 + (16 * wave + cc) * INP + 16 * kb0 + 4 * g;
    f32x4 m1[SKQ], v1[SKQ];
#pragma unroll
    for (int q = 0; q < SKQ; ++q) {
      f32x4 w = {0.f, 0.f, 0.f, 0.f};
      m1[q] = w;
      v1[q] = w;
      if (q < nkb) {  // block-uniform
        w = *reinterpret_cast<const f32x4*>(gp + own_glb + 16 * q);
        m1[q] = *reinterpret_cast<const f32x4*>(gm + own_glb + 16 * q);
        v1[q] = *reinterpret_cast<const f32x4*>(gv + own_glb + 16 * q);
      }
      *reinterpret_cast<f32x4*>(sW + own_lds + 16 * q) = w;
    }
    {
      const int step0 = a.adam_step[net];
      const float lr = critic ? a.hp.lr_critic : a.hp.lr_actor;
      for (int k = tid; k < a.nmb; k += 256) {
        const double t = (double)(step0 + k + 1);
        sBC[2 * k + 0] = (float)((double)lr / (1.0 - pow((double)beta1, t)));
        sBC[2 * k + 1] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, t)));
      }
    }
    const int c = 16 * wave + j;  // this lane's sample column
    long row_nxt;
    {
      const int B0 = (int)min((long)a.B, a.M);
      const long p0 = (c < B0) ? c : 0;
      row_nxt = perm_p ? perm_p[p0] : p0;
    }
    f32x4 xr[SKQ];
    {
      const float* __restrict__ xrow = obs_p + row_nxt * ld_obs;
#pragma unroll
      for (int q = 0; q < SKQ; ++q) xr[q] = load_chunk(xrow, 16 * (kb0 + q) + 4 * g);
    }
    __syncthreads();
    for (int mb = 0; mb < a.nmb; ++mb) {
      const long mb_lo = (long)mb * a.B;
      const bool have_next = mb + 1 < a.nmb;
      // the permutation entry of the NEXT step's sample: requested now, its row is gathered after this step's
      // forward pass (two dependent memory round trips that would otherwise open every step)
      long row_nn = row_nxt;
      if (have_next) {
        const long nlo = mb_lo + a.B;
        const int nB = (int)(min(nlo + a.B, a.M) - nlo);
        const long np = nlo + ((c < nB) ? c : 0);
        row_nn = perm_p ? perm_p[np] : np;
      }
      // ---- this step's columns: B operand of the forward product, and (transposed) A operand of dW1
      f32x4 x[SKQ];
#pragma unroll
      for (int q = 0; q < SKQ; ++q) {
        x[q] = mask_chunk(xr[q], 16 * (kb0 + q) + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) sX[(16 * q + 4 * g + r) * SSLD + c] = x[q][r];
      }
      // ---- partial pre-activation: features 16 t + 4 g + r of sample c
      f32x4 acc[HT];
#pragma unroll
      for (int t = 0; t < HT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // (all SKQ blocks unconditionally: blocks past the slice hold zero weights and meet zero-masked columns -- a
      // block-uniform branch around each group is a scheduling barrier: measured 18 k instead of 5 k cycles in the
      // dW1 loop below)
#pragma unroll
      for (int q = 0; q < SKQ; ++q) {
        f32x4 w[HT];
#pragma unroll
        for (int t = 0; t < HT; ++t) w[t] = *reinterpret_cast<const f32x4*>(sW + (16 * t + i) * SXLD + 16 * q + 4 * g);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < HT; ++t) acc[t] = OSA_MFMA(w[t][s], x[q][s], acc[t]);
      }
      {
        f32x4* dst = reinterpret_cast<f32x4*>(xn + SX_PART + hc * 4096);
#pragma unroll
        for (int t = 0; t < HT; ++t) dst[t * 256 + tid] = acc[t];
      }
      STICK(0);
      osa_xch_release();
      osa_lds_barrier();  // partials performed; sX complete
      if (tid == 0) __hip_atomic_store(flags + SF_PART + hc, mb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      STICK(1);
      // ---- gather the next step's columns while the leader works
      {
        const float* __restrict__ xrow = obs_p + row_nn * ld_obs;
#pragma unroll
        for (int q = 0; q < SKQ; ++q) xr[q] = load_chunk(xrow, 16 * (kb0 + q) + 4 * g);
      }
      row_nxt = row_nn;
      // the lane's own weights (norm share, Adam)
      f32x4 wv[SKQ];
#pragma unroll
      for (int q = 0; q < SKQ; ++q) wv[q] = *reinterpret_cast<const f32x4*>(sW + own_lds + 16 * q);
      // ---- dz1 of this step
      // The data carries its own readiness: the leader resets the buffer of the OTHER step parity to all-ones words
      // one step ahead (and the host both buffers before the launch), so a lane polls ITS sixteen words with
      // cache-bypassing loads until none of them is the sentinel -- one round trip for the hand-off, and nothing
      // that relies on the order in which two loads are performed.  (Rounds 2-3 polled a flag and requested the
      // data right behind it in the same trip: loads RETURN in issue order, but the flag and the data live in
      // different memory channels and the data load can be PERFORMED first -- stale dz1 of an older step in about 1
      // of 1000 passes at 8 virtual ranks, tools/dp_stress.py.)
      f32x4 a1[4];
      {
        const float* dz = xn + SX_DZ + (mb & 1) * 4096 + (16 * wave + i) * 64 + 4 * g;
        int spins = 0;
        while (!dead) {
#pragma unroll
          for (int sb = 0; sb < 4; ++sb) a1[sb] = osa_load_bypass4(dz + 16 * sb);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          unsigned m = 0u;
#pragma unroll
          for (int sb = 0; sb < 4; ++sb) {
            m = osa_max3u(m, __float_as_uint(a1[sb][0]), __float_as_uint(a1[sb][1]));
            m = osa_max3u(m, __float_as_uint(a1[sb][2]), __float_as_uint(a1[sb][3]));
          }
          if (__builtin_amdgcn_ballot_w64(m == SX_SENT) == 0) break;
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 20)) {
            __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dead = true;
          }
        }
      }
      STICK(2);
      // ---- dW1[f][k] = sum_s dz1[f][s] x[s][k], D[i = input 4g + r][j = feature cc]
      f32x4 gq[SKQ];
#pragma unroll
      for (int q = 0; q < SKQ; ++q) gq[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        f32x4 b[SKQ];
#pragma unroll
        for (int q = 0; q < SKQ; ++q) b[q] = *reinterpret_cast<const f32x4*>(sX + (16 * q + i) * SSLD + 16 * sb + 4 * g);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int q = 0; q < SKQ; ++q) gq[q] = OSA_MFMA(b[q][s], a1[sb][s], gq[q]);
      }
      STICK(3);
      f32x4 acc_g = {0.f, 0.f, 0.f, 0.f}, acc_p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < SKQ; ++q) {
        if (l2) gq[q] = gq[q] + wv[q] * c2;
        acc_g = acc_g + gq[q] * gq[q];
        acc_p = acc_p + wv[q] * wv[q];
      }
      float gsq = (acc_g.x + acc_g.y) + (acc_g.z + acc_g.w);
      float psq = (acc_p.x + acc_p.y) + (acc_p.z + acc_p.w);
      gsq = osa_wave_sum_dpp(gsq);
      psq = osa_wave_sum_dpp(psq);
      if (lane == 0) {
        red[2 * wave + 0] = gsq;
        red[2 * wave + 1] = psq;
      }
      osa_lds_barrier();
      if (tid == 0) {
        osa_slot_put(slots + 2 * hc, (red[0] + red[2]) + (red[4] + red[6]), mb + 1);
        osa_slot_put(slots + 2 * hc + 1, (red[1] + red[3]) + (red[5] + red[7]), mb + 1);
      }
      STICK(4);
      // ---- all norm shares (lane k <= C polls share k; summed in slot order by everybody)
      if (tid <= C) red[8 + tid] = osa_slot_wait(slots + 2 * tid, mb + 1, err, dead);
      osa_lds_barrier();
      float t_gsq = 0.f;
      for (int k = 0; k <= C; ++k) t_gsq += red[8 + k];
      const float step_size = sBC[2 * mb], inv_bc2_sqrt = sBC[2 * mb + 1];
      float gscale = 1.f;
      if (a.hp.use_max_grad_norm) {
        gscale = a.hp.max_grad_norm / (sqrtf(t_gsq) + 1e-6f);
        gscale = gscale > 1.f ? 1.f : gscale;
      }
      STICK(5);
      if constexpr (DP) {  // average of the ranks' locally clipped slices (clip-then-average)
        float nob = 0.f;
#ifdef OSA_SPLIT_CLOCKS
        osa_split_dp_average<SKQ>(gq, nob, gscale, nullptr, dp_slabs(mb), rk, W, dp_cnt, W * (mb + 1), tid, err, dead, dclk);
#else
        osa_split_dp_average<SKQ>(gq, nob, gscale, nullptr, dp_slabs(mb), rk, W, dp_cnt, W * (mb + 1), tid, err, dead);
#endif
        gscale = 1.f;
      }
#pragma unroll
      for (int q = 0; q < SKQ; ++q) {
        const f32x4 w = osa_adam_update4(gq[q] * gscale, m1[q], v1[q], wv[q], beta1, beta2, step_size, inv_bc2_sqrt, aeps);
        *reinterpret_cast<f32x4*>(sW + own_lds + 16 * q) = w;
      }
      osa_lds_barrier();  // the slice is consistent; sX and red are free
      STICK(6);
    }
    // ---- write back the slice and its moments (data-parallel form: the replicas are identical, rank 0 writes)
    if (DP && rk != 0) return;
#pragma unroll
    for (int q = 0; q < SKQ; ++q) {
      if (q < nkb) {
        *reinterpret_cast<f32x4*>(gp + own_glb + 16 * q) = *reinterpret_cast<const f32x4*>(sW + own_lds + 16 * q);
        *reinterpret_cast<f32x4*>(gm + own_glb + 16 * q) = m1[q];
        *reinterpret_cast<f32x4*>(gv + own_glb + 16 * q) = v1[q];
      }
    }
#ifdef OSA_SPLIT_CLOCKS
    if (tid == 0 && hc == 0 && rk == 0 && a.nmb >= 8) {  // helper 0 of network `net` -> row net (long since consumed), columns 8..15
      for (int k = 0; k < 8; ++k) a.stats[(long)net * SNSTAT + 8 + k] = (float)sdbg[k] / (float)a.nmb;
      for (int k = 0; k < 3; ++k) a.stats[(long)(3 + net) * SNSTAT + 8 + k] = (float)dclk[k] / (float)a.nmb;
    }
#endif
    return;
  }

  // =======================================================================================================
  // leader: layers 2 and 3, loss, backward pass, dW2 / dW3 / biases
  // =======================================================================================================
  float* sW2 = smem;                      // [H][SSLD]
  float* sW3 = sW2 + H * SSLD;            // [OUTP][SSLD]
  float* sB1 = sW3 + OUTP * SSLD;         // [H]
  float* sB2 = sB1 + H;                   // [H]
  float* sB3 = sB2 + H;                   // [OUTP]
  float* sLS = sB3 + OUTP;                // [OUTP]
  float* sH1 = sLS + OUTP;                // [H][SSLD]  element (feature f, sample c)
  float* sH2 = sH1 + H * SSLD;
  float* sZ1 = sH2 + H * SSLD;
  float* sZ2 = sZ1 + H * SSLD;
  float* sDO = sZ2 + H * SSLD;            // [OUTP][SSLD]
  float* sDL = sDO + OUTP * SSLD;         // [OUTP][SSLD]
  float* red = sDL + OUTP * SSLD;         // [64]
  // transposed copies for the backward pass (its A operands W^T[16t+i][k .. k+3] are then ONE 16-byte LDS read
  // instead of four scalar ones): kept in step by the Adam writers
  constexpr int S3LD = OUTP + 4;
  float* sW2T = red + 64;                 // [H][SSLD]   element (input feature f, output feature o) = W2[o][f]
  float* sW3T = sW2T + H * SSLD;          // [H][S3LD]   element (input feature f, output o) = W3[o][f]
  // per-dimension constants of the Gaussian log-density, refreshed by the thread that owns the log_std element
  // whenever Adam changes it (every lane would otherwise evaluate exp / log / a division per element and step)
  float* sVAR = sW3T + H * S3LD;          // [OUTP]  sigma^2 = exp(log_std)^2
  float* sIV = sVAR + OUTP;               // [OUTP]  1 / sigma^2
  float* sLC = sIV + OUTP;                // [OUTP]  log(sigma) (the reference evaluates scale.log())
  const bool leader = tid == 192;
  for (int e = tid; e < H * H; e += 256) sW2[(e >> 6) * SSLD + (e & 63)] = gp[nd.oW2 + e];
  for (int e = tid; e < OUTP * H; e += 256) sW3[(e >> 6) * SSLD + (e & 63)] = gp[nd.oW3 + e];
  for (int e = tid; e < H * H; e += 256) sW2T[(e & 63) * SSLD + (e >> 6)] = gp[nd.oW2 + e];
  for (int e = tid; e < OUTP * H; e += 256) sW3T[(e & 63) * S3LD + (e >> 6)] = gp[nd.oW3 + e];
  if (tid < H) {
    sB1[tid] = gp[nd.ob1 + tid];
    sB2[tid] = gp[nd.ob2 + tid];
  }
  if (tid < OUTP) {
    sB3[tid] = gp[nd.ob3 + tid];
    const float ls = gp[nd.oLS + tid];
    sLS[tid] = ls;
    const float sd = expf(ls);
    sVAR[tid] = sd * sd;
    sIV[tid] = 1.f / (sd * sd);
    sLC[tid] = logf(sd);
  }
  // ownership (as osa_ppo_pass_kernel): W2[(16w+4g+r)][16ti+cc], W3[(16o+4g+r)][16w+cc], one bias-like
  // scalar per thread; Adam moments in registers
  float mb_ = 0.f, vb_ = 0.f;
  int boff = -1;
  float* sbias = sB1;
  if (tid < H) { boff = nd.ob1 + tid; sbias = sB1 + tid; }
  else if (tid < 2 * H) { boff = nd.ob2 + tid - H; sbias = sB2 + tid - H; }
  else if (tid < 2 * H + OUTP) { boff = nd.ob3 + tid - 2 * H; sbias = sB3 + tid - 2 * H; }
  else if (tid < 2 * H + 2 * OUTP) { boff = nd.oLS + tid - 2 * H - OUTP; sbias = sLS + tid - 2 * H - OUTP; }
  if (critic && boff >= nd.oLS) boff = -1;  // critics have no log_std
  if (boff >= 0) {
    mb_ = gm[boff];
    vb_ = gv[boff];
  }
  f32x4 m2[HT], v2[HT], m3[OT], v3[OT];
#pragma unroll
  for (int ti = 0; ti < HT; ++ti)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int off = nd.oW2 + (16 * wave + 4 * g + r) * H + 16 * ti + cc;
      m2[ti][r] = gm[off];
      v2[ti][r] = gv[off];
    }
#pragma unroll
  for (int o = 0; o < OT; ++o)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int off = nd.oW3 + (16 * o + 4 * g + r) * H + 16 * wave + cc;
      m3[o][r] = gm[off];
      v3[o][r] = gv[off];
    }
  const int step0 = a.adam_step[net];
  const float lr = critic ? a.hp.lr_critic : a.hp.lr_actor;
  // Adam's bias corrections of every step of the launch, tabulated once (float64 like torch): columns
  // 10 + 2 net, 11 + 2 net of the step's statistics row
  for (int k = tid; k < a.nmb; k += 256) {
    const double t = (double)(step0 + k + 1);
    float* row = a.stats + (long)k * SNSTAT;
    row[10 + 2 * net] = (float)((double)lr / (1.0 - pow((double)beta1, t)));
    row[11 + 2 * net] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, t)));
  }
  float lam = 0.f;
  if (net == 0 && a.lagrange) lam = *a.lagrange;
  const float* __restrict__ tgt = ((net == 1) ? a.tgt_r : a.tgt_c) + roff;
  __syncthreads();  // LDS master copy + bias-correction table complete

#define SPUT_TILE(S, V, T)                                                           \
  _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) (S)[(16 * (T) + 4 * g + r_) * SSLD + c] = (V)[r_]

  const int c = 16 * wave + j;  // this lane's sample column
  long row_nxt;
  {
    const int B0 = (int)min((long)a.B, a.M);
    const long p0 = (c < B0) ? c : 0;
    row_nxt = perm_p ? perm_p[p0] : p0;
  }
  // per-sample scalars of the step, gathered one step ahead
  float n_act[4 * OT], n_logp = 0.f, n_advr = 0.f, n_advc = 0.f, n_tgt = 0.f;
  auto gather = [&](long row) {
    if (net == 0) {  // block-uniform
      const float* __restrict__ arow = act_p + row * a.ld_act;
#pragma unroll
      for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int r = 0; r < 4; ++r) n_act[4 * o + r] = arow[min(16 * o + 4 * g + r, nd.act_dim - 1)];
      n_logp = logp_p[row];
      n_advr = advr_p[row];
      n_advc = advc_p[row];
    } else {
      n_tgt = tgt[row];
    }
  };
#pragma unroll
  for (int k = 0; k < 4 * OT; ++k) n_act[k] = 0.f;
  gather(row_nxt);
  for (int mb = 0; mb < a.nmb; ++mb) {
    const long mb_lo = (long)mb * a.B;
    const int Bcur = (int)(min(mb_lo + a.B, a.M) - mb_lo);
    const float invB = 1.f / (float)Bcur;
    const bool valid = c < Bcur;
    const bool have_next = mb + 1 < a.nmb;
    long row_nn = row_nxt;
    if (have_next) {
      const long nlo = mb_lo + a.B;
      const int nB = (int)(min(nlo + a.B, a.M) - nlo);
      const long np = nlo + ((c < nB) ? c : 0);
      row_nn = perm_p ? perm_p[np] : np;
    }
    float s_act[4 * OT];
#pragma unroll
    for (int k = 0; k < 4 * OT; ++k) s_act[k] = n_act[k];
    const float s_logp = n_logp, s_advr = n_advr, s_advc = n_advc, s_tgt = n_tgt;
    const float* bc_row = a.stats + (long)mb * SNSTAT + 10 + 2 * net;
    const float step_size = bc_row[0], inv_bc2_sqrt = bc_row[1];
    float ent_pre = 0.f;
    if (net == 0 && leader) {  // entropy of the pre-update policy
      for (int d = 0; d < nd.act_dim; ++d) ent_pre += 1.41893853320467274178f + sLS[d];
      ent_pre /= (float)nd.act_dim;
    }
    // ================= forward =================
    f32x4 h1[HT], h2[HT], out[OT];
#pragma unroll
    for (int t = 0; t < HT; ++t) h1[t] = *reinterpret_cast<const f32x4*>(sB1 + 16 * t + 4 * g);
    osa_xch_wait(flags + SF_PART, C, mb + 1, tid, err, dead);
    osa_lds_barrier();
    osa_xch_acquire();
    STICK(0);
    {
      // every helper has sent its partials of this step, i.e. has consumed dz1 of the previous one: its buffer (the
      // other step parity) goes back to "not published" for the step after this one.  Performed, with the dz1 of
      // this step, before this workgroup's norm share -- which every helper reads before it polls that buffer.
      const float sf = __uint_as_float(SX_SENT);
      f32x4* nx = reinterpret_cast<f32x4*>(xn + SX_DZ + ((mb + 1) & 1) * 4096);
#pragma unroll
      for (int k = 0; k < 4; ++k) nx[k * 256 + tid] = (f32x4){sf, sf, sf, sf};
    }
    {
      // all C x 4 loads in flight together (one round trip to the device-coherent level, not C)
      const f32x4* src = reinterpret_cast<const f32x4*>(xn + SX_PART);
      f32x4 pt[SCMAX][HT];
#pragma unroll
      for (int k = 0; k < SCMAX; ++k) {
        if (k < C) {  // block-uniform
#pragma unroll
          for (int t = 0; t < HT; ++t) pt[k][t] = src[k * 1024 + t * 256 + tid];
        }
      }
#pragma unroll
      for (int k = 0; k < SCMAX; ++k) {
        if (k < C) {
#pragma unroll
          for (int t = 0; t < HT; ++t) h1[t] = h1[t] + pt[k][t];
        }
      }
    }
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      h1[t] = osa_tanh4(h1[t]);
      SPUT_TILE(sH1, h1[t], t);
    }
#pragma unroll
    for (int t = 0; t < HT; ++t) h2[t] = *reinterpret_cast<const f32x4*>(sB2 + 16 * t + 4 * g);
    {
      // (LDS fragment reads one K block ahead of the MFMAs that consume them: with one wave per SIMD nothing else
      // hides an LDS round trip)
      f32x4 wn[HT];
#pragma unroll
      for (int t = 0; t < HT; ++t) wn[t] = *reinterpret_cast<const f32x4*>(sW2 + (16 * t + i) * SSLD + 4 * g);
#pragma unroll
      for (int kb = 0; kb < HT; ++kb) {
        f32x4 w[HT];
#pragma unroll
        for (int t = 0; t < HT; ++t) w[t] = wn[t];
        if (kb + 1 < HT) {
#pragma unroll
          for (int t = 0; t < HT; ++t) wn[t] = *reinterpret_cast<const f32x4*>(sW2 + (16 * t + i) * SSLD + 16 * (kb + 1) + 4 * g);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < HT; ++t) h2[t] = OSA_MFMA(w[t][s], h1[kb][s], h2[t]);
      }
    }
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      h2[t] = osa_tanh4(h2[t]);
      SPUT_TILE(sH2, h2[t], t);
    }
#pragma unroll
    for (int o = 0; o < OT; ++o) out[o] = *reinterpret_cast<const f32x4*>(sB3 + 16 * o + 4 * g);
    {
      f32x4 w3f[HT][OT];
#pragma unroll
      for (int kb = 0; kb < HT; ++kb)
#pragma unroll
        for (int o = 0; o < OT; ++o) w3f[kb][o] = *reinterpret_cast<const f32x4*>(sW3 + (16 * o + i) * SSLD + 16 * kb + 4 * g);
#pragma unroll
      for (int kb = 0; kb < HT; ++kb)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int o = 0; o < OT; ++o) out[o] = OSA_MFMA(w3f[kb][o][s], h2[kb][s], out[o]);
    }
    // ================= loss, dL/d(out) (osa_mb_grad_kernel's code path without extensions) =================
    STICK(1);
    f32x4 dO[OT], dLS[OT];
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      dO[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dLS[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float loss_part = 0.f, ratio_part = 0.f;
    if (net == 0) {
      float lp = 0.f;
      f32x4 zv[OT], ivar[OT];
#pragma unroll
      for (int o = 0; o < OT; ++o) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = 16 * o + 4 * g + r;
          zv[o][r] = 0.f;
          ivar[o][r] = 0.f;
          if (d < nd.act_dim && valid) {
            const float z = s_act[4 * o + r] - out[o][r];
            zv[o][r] = z;
            const float iv = sIV[d];
            ivar[o][r] = iv;
            // (z^2 / (2 sigma^2) as a product with the tabulated 1 / sigma^2, as osa_ppo_pass_kernel does: a division per
            // element and lane is ~12 VALU instructions on the leader's critical path)
            lp += -(z * z) * (0.5f * iv) - sLC[d] - 0.91893853320467274178f;
          }
        }
      }
      lp = osa_sum_over_groups(lp);
      const float ratio = valid ? __builtin_amdgcn_exp2f((lp - s_logp) * 1.44269504088896340736f) : 0.f;
      if (valid) {
        const float adv = (s_advr - lam * s_advc) / (1.f + lam);  // ppo_lag.py:101-102
        float dratio, li;
        if (a.loss_kind == 0) {  // base/ppo.py:66-78
          const float lo = 1.f - a.hp.clip, hi = 1.f + a.hp.clip;
          const float rc = fminf(fmaxf(ratio, lo), hi);
          const float s1 = ratio * adv, s2 = rc * adv;
          const bool inrange = ratio >= lo && ratio <= hi;
          li = -fminf(s1, s2);
          dratio = (s1 < s2 || inrange) ? -adv : 0.f;
        } else {  // policy_gradient.py:574-578
          li = -(ratio * adv);
          dratio = -adv;
        }
        const float dlogp = dratio * ratio * invB;
        if (g == 0) {
          loss_part = li;
          ratio_part = ratio;
        }
#pragma unroll
        for (int o = 0; o < OT; ++o) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z = zv[o][r], iv = ivar[o][r];
            dO[o][r] = dlogp * z * iv;
            dLS[o][r] = (iv != 0.f) ? dlogp * (z * z * iv - 1.f) : 0.f;
          }
        }
      }
    } else if (valid) {
      const float diff = out[0][0] - s_tgt;
      if (g == 0) {
        loss_part = diff * diff;
        dO[0][0] = 2.f * diff * invB;
      }
    }
    // ================= backward through the hidden layers (S layout) =================
    f32x4 z2[HT], z1[HT];
    {
      f32x4 wt[HT][OT], acc[HT];
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < OT; ++o)  // A[i][k] = W3^T[16t+i][16o+4g+s]
          wt[t][o] = *reinterpret_cast<const f32x4*>(sW3T + (16 * t + i) * S3LD + 16 * o + 4 * g);
      }
#pragma unroll
      for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < HT; ++t) acc[t] = OSA_MFMA(wt[t][o][s], dO[o][s], acc[t]);  // four independent chains
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        z2[t] = acc[t] * (1.f - h2[t] * h2[t]);
        SPUT_TILE(sZ2, z2[t], t);
      }
    }
    {
      float* dz = xn + SX_DZ + (mb & 1) * 4096;
      f32x4 wn[HT];  // A[i][k] = W2^T[16t+i][16kb+4g+s]: the next tile's four fragments one tile ahead
#pragma unroll
      for (int kb = 0; kb < HT; ++kb) wn[kb] = *reinterpret_cast<const f32x4*>(sW2T + i * SSLD + 16 * kb + 4 * g);
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        f32x4 w[HT];
#pragma unroll
        for (int kb = 0; kb < HT; ++kb) w[kb] = wn[kb];
        if (t + 1 < HT) {
#pragma unroll
          for (int kb = 0; kb < HT; ++kb)
            wn[kb] = *reinterpret_cast<const f32x4*>(sW2T + (16 * (t + 1) + i) * SSLD + 16 * kb + 4 * g);
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < HT; ++kb)
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = OSA_MFMA(w[kb][s], z2[kb][s], acc);
        z1[t] = acc * (1.f - h1[t] * h1[t]);
        // dz1 leaves for the helpers as soon as a tile is finished: row = feature, 16 consecutive samples per
        // (g, r) = 64-byte segments
#pragma unroll
        for (int r = 0; r < 4; ++r) dz[(16 * t + 4 * g + r) * 64 + c] = z1[t][r];
        SPUT_TILE(sZ1, z1[t], t);
      }
    }
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      SPUT_TILE(sDO, dO[o], o);
      SPUT_TILE(sDL, dLS[o], o);
    }
    STICK(2);
    osa_xch_release();
    osa_lds_barrier();  // (A) dz1 (and the other buffer's reset) performed, tiles complete
    STICK(3);
    // ================= weight gradients (registers) =================
    f32x4 g2[HT], g3[OT];
    {
      f32x4 a2[4];
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) a2[sb] = *reinterpret_cast<const f32x4*>(sZ2 + (16 * wave + i) * SSLD + 16 * sb + 4 * g);
#pragma unroll
      for (int ti = 0; ti < HT; ++ti) g2[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // (fragments one sample block ahead of their MFMAs; the output layer's fragments all up front and its OT
      // accumulator chains interleaved)
      f32x4 bn[HT];
#pragma unroll
      for (int ti = 0; ti < HT; ++ti) bn[ti] = *reinterpret_cast<const f32x4*>(sH1 + (16 * ti + i) * SSLD + 4 * g);
      f32x4 av3[OT][4], b3[4];
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        b3[sb] = *reinterpret_cast<const f32x4*>(sH2 + (16 * wave + i) * SSLD + 16 * sb + 4 * g);
#pragma unroll
        for (int o = 0; o < OT; ++o)
          av3[o][sb] = *reinterpret_cast<const f32x4*>(sDO + (16 * o + i) * SSLD + 16 * sb + 4 * g);
      }
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        f32x4 b[HT];
#pragma unroll
        for (int ti = 0; ti < HT; ++ti) b[ti] = bn[ti];
        if (sb + 1 < 4) {
#pragma unroll
          for (int ti = 0; ti < HT; ++ti)
            bn[ti] = *reinterpret_cast<const f32x4*>(sH1 + (16 * ti + i) * SSLD + 16 * (sb + 1) + 4 * g);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int ti = 0; ti < HT; ++ti) g2[ti] = OSA_MFMA(a2[sb][s], b[ti][s], g2[ti]);
      }
#pragma unroll
      for (int o = 0; o < OT; ++o) g3[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sb = 0; sb < 4; ++sb)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int o = 0; o < OT; ++o) g3[o] = OSA_MFMA(av3[o][sb][s], b3[sb][s], g3[o]);
    }
    // bias-like gradient owned by this thread: row sum over the 64 samples
    float gb = 0.f;
    {
      const float* srow = (tid < H) ? sZ1 + tid * SSLD
                          : (tid < 2 * H) ? sZ2 + (tid - H) * SSLD
                          : (tid < 2 * H + OUTP) ? sDO + (tid - 2 * H) * SSLD
                          : (tid < 2 * H + 2 * OUTP) ? sDL + (tid - 2 * H - OUTP) * SSLD
                                                     : sZ1;
      float rs = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(srow + 4 * k);
        rs += q.x;
        rs += q.y;
        rs += q.z;
        rs += q.w;
      }
      gb = (boff >= 0) ? rs : 0.f;
    }
    if (boff >= 0 && net == 0 && boff >= nd.oLS && (boff - nd.oLS) < nd.act_dim && a.hp.entropy_coef != 0.f)
      gb -= a.hp.entropy_coef / (float)nd.act_dim;
    // ================= + 2 coef w (critics), squared norms =================
    f32x4 w2r[HT], w3r[OT];
#pragma unroll
    for (int ti = 0; ti < HT; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) w2r[ti][r] = sW2[(16 * wave + 4 * g + r) * SSLD + 16 * ti + cc];
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r) w3r[o][r] = sW3[(16 * o + 4 * g + r) * SSLD + 16 * wave + cc];
    float wb = *sbias;
    wb = (boff >= 0) ? wb : 0.f;
    f32x4 acc_g = {0.f, 0.f, 0.f, 0.f}, acc_p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ti = 0; ti < HT; ++ti) {
      const f32x4 w = w2r[ti];
      if (l2) g2[ti] = g2[ti] + w * c2;
      acc_p = acc_p + w * w;
      acc_g = acc_g + g2[ti] * g2[ti];
    }
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      const f32x4 w = w3r[o];
      if (l2) g3[o] = g3[o] + w * c2;
      acc_p = acc_p + w * w;
      acc_g = acc_g + g3[o] * g3[o];
    }
    float gsq = (acc_g.x + acc_g.y) + (acc_g.z + acc_g.w);
    float psq = (acc_p.x + acc_p.y) + (acc_p.z + acc_p.w);
    if (boff >= 0) {
      if (l2) gb += c2 * wb;
      psq += wb * wb;
      gsq += gb * gb;
    }
    gsq = osa_wave_sum_dpp(gsq);
    psq = osa_wave_sum_dpp(psq);
    loss_part = osa_wave_sum_dpp(loss_part);
    ratio_part = osa_wave_sum_dpp(ratio_part);
    if (lane == 0) {
      red[4 * wave + 0] = gsq;
      red[4 * wave + 1] = psq;
      red[4 * wave + 2] = loss_part;
      red[4 * wave + 3] = ratio_part;
    }
    STICK(4);
    osa_lds_barrier();  // (B)
    if (tid == 0) {
      osa_slot_put(slots + 2 * C, red[0] + red[4] + red[8] + red[12], mb + 1);
      osa_slot_put(slots + 2 * C + 1, red[1] + red[5] + red[9] + red[13], mb + 1);
    }
    const float t_loss = red[2] + red[6] + red[10] + red[14];
    const float t_ratio = red[3] + red[7] + red[11] + red[15];
    STICK(5);
    if (tid <= C) red[16 + tid] = osa_slot_wait(slots + 2 * tid, mb + 1, err, dead);
    else if (tid >= 64 && tid - 64 <= C) red[24 + tid - 64] = osa_slot_wait(slots + 2 * (tid - 64) + 1, mb + 1, err, dead);
    osa_lds_barrier();
    float t_gsq = 0.f, t_psq = 0.f;
    for (int k = 0; k <= C; ++k) {
      t_gsq += red[16 + k];
      t_psq += red[24 + k];
    }
    const float total_norm = sqrtf(t_gsq);
    float gscale = 1.f;
    if (a.hp.use_max_grad_norm) {
      gscale = a.hp.max_grad_norm / (total_norm + 1e-6f);
      gscale = gscale > 1.f ? 1.f : gscale;
    }
    STICK(6);
    float st_loss = t_loss * invB - ((net == 0) ? a.hp.entropy_coef * ent_pre : 0.f), st_ratio = t_ratio * invB,
          st_psq = t_psq, st_norm = total_norm, st_ent = ent_pre;
    if constexpr (DP) {
      // average of the ranks' locally clipped (W2, W3, bias-like) gradients; the step's statistics travel in the
      // slab's tail and are averaged by rank 0's leader (what Logger.get_stats averages across ranks)
      f32x4 gg[HT + OT];
#pragma unroll
      for (int ti = 0; ti < HT; ++ti) gg[ti] = g2[ti];
#pragma unroll
      for (int o = 0; o < OT; ++o) gg[HT + o] = g3[o];
      float* t5 = red + 40;  // (red: 64 floats; 16 .. 31 hold the norm shares until barrier C)
      if (leader) { t5[0] = st_loss; t5[1] = st_ratio; t5[2] = st_psq; t5[3] = st_norm; t5[4] = st_ent; }
      osa_lds_barrier();
      float* slabs = dp_slabs(mb);
#ifdef OSA_SPLIT_CLOCKS
      osa_split_dp_average<HT + OT>(gg, gb, gscale, t5, slabs, rk, W, dp_cnt, W * (mb + 1), tid, err, dead, dclk);
#else
      osa_split_dp_average<HT + OT>(gg, gb, gscale, t5, slabs, rk, W, dp_cnt, W * (mb + 1), tid, err, dead);
#endif
#pragma unroll
      for (int ti = 0; ti < HT; ++ti) g2[ti] = gg[ti];
#pragma unroll
      for (int o = 0; o < OT; ++o) g3[o] = gg[HT + o];
      gscale = 1.f;
      if (leader && rk == 0) {
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < W; ++r) {
          const float* t = slabs + (long)r * SDPW + SKQ * 1024 + 256 + 1;
          for (int k = 0; k < 5; ++k) acc[k] += t[k];
        }
        const float invW = 1.f / (float)W;
        st_loss = acc[0] * invW; st_ratio = acc[1] * invW; st_psq = acc[2] * invW; st_norm = acc[3] * invW;
        st_ent = acc[4] * invW;
      }
    }
    // ================= Adam =================
#pragma unroll
    for (int ti = 0; ti < HT; ++ti) {
      const f32x4 w = osa_adam_update4(g2[ti] * gscale, m2[ti], v2[ti], w2r[ti], beta1, beta2, step_size, inv_bc2_sqrt, aeps);
#pragma unroll
      for (int r = 0; r < 4; ++r) sW2[(16 * wave + 4 * g + r) * SSLD + 16 * ti + cc] = w[r];
      *reinterpret_cast<f32x4*>(sW2T + (16 * ti + cc) * SSLD + 16 * wave + 4 * g) = w;
    }
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      const f32x4 w = osa_adam_update4(g3[o] * gscale, m3[o], v3[o], w3r[o], beta1, beta2, step_size, inv_bc2_sqrt, aeps);
#pragma unroll
      for (int r = 0; r < 4; ++r) sW3[(16 * o + 4 * g + r) * SSLD + 16 * wave + cc] = w[r];
      *reinterpret_cast<f32x4*>(sW3T + (16 * wave + cc) * S3LD + 16 * o + 4 * g) = w;
    }
    if (boff >= 0) {
      float mv_ = mb_, vv_ = vb_;
      const float nw = osa_adam_update(gb * gscale, mv_, vv_, wb, beta1, beta2, step_size, inv_bc2_sqrt, aeps);
      *sbias = nw;
      mb_ = mv_;
      vb_ = vv_;
      if (boff >= nd.oLS) {  // (actor only) the log-density constants of the new log_std
        const float sd = expf(nw);
        sVAR[boff - nd.oLS] = sd * sd;
        sIV[boff - nd.oLS] = 1.f / (sd * sd);
        sLC[boff - nd.oLS] = logf(sd);
      }
    }
    if (leader && rk == 0) {
      float* st = a.stats + (long)mb * SNSTAT;
      if (net == 0) {
        st[2] = st_loss;
        st[3] = st_ratio;
        st[4] = st_ent;
        st[7] = st_norm;
      } else {
        st[net - 1] = st_loss;
        st[4 + net] = st_psq;
        st[7 + net] = st_norm;
      }
    }
    // ---- the next step's scalars: requested here, where the leader is about to wait for the helpers' partials
    // anyway (between the dz1 hand-off and the norm shares it is on the critical path: the actor's 11 gathers with
    // their address arithmetic cost 1.2 k cycles there)
    gather(row_nn);
    row_nxt = row_nn;
    osa_lds_barrier();  // (C) frees tiles and `red`; the LDS master copy is consistent
    STICK(7);
  }
#undef SPUT_TILE
#ifdef OSA_SPLIT_CLOCKS
  if (tid == 0 && rk == 0 && a.nmb >= 8) {  // leader of network `net` -> row net, columns 0..7: mean cycles per step
    for (int k = 0; k < 8; ++k) a.stats[(long)net * SNSTAT + k] = (float)sdbg[k] / (float)a.nmb;
    for (int k = 0; k < 3; ++k) a.stats[(long)(3 + net) * SNSTAT + k] = (float)dclk[k] / (float)a.nmb;
  }
#endif
  // ---- write back: LDS master copy, Adam state (data-parallel form: the replicas are identical, rank 0 writes)
  if (DP && rk != 0) return;
  for (int e = tid; e < H * H; e += 256) gp[nd.oW2 + e] = sW2[(e >> 6) * SSLD + (e & 63)];
  for (int e = tid; e < OUTP * H; e += 256) gp[nd.oW3 + e] = sW3[(e >> 6) * SSLD + (e & 63)];
  if (tid < H) {
    gp[nd.ob1 + tid] = sB1[tid];
    gp[nd.ob2 + tid] = sB2[tid];
  }
  if (tid < OUTP) {
    gp[nd.ob3 + tid] = sB3[tid];
    if (!critic) gp[nd.oLS + tid] = sLS[tid];
  }
#pragma unroll
  for (int ti = 0; ti < HT; ++ti)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int off = nd.oW2 + (16 * wave + 4 * g + r) * H + 16 * ti + cc;
      gm[off] = m2[ti][r];
      gv[off] = v2[ti][r];
    }
#pragma unroll
  for (int o = 0; o < OT; ++o)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int off = nd.oW3 + (16 * o + 4 * g + r) * H + 16 * wave + cc;
      gm[off] = m3[o][r];
      gv[off] = v3[o][r];
    }
  if (boff >= 0) {
    gm[boff] = mb_;
    gv[boff] = vb_;
  }
  if (tid == 0) a.adam_step[net] = step0 + a.nmb;
}

static size_t osa_split_lds_bytes(int OT, int nmb) {
  const int H = 64, OUTP = 16 * OT;
  const size_t lead = (size_t)H * SSLD + (size_t)OUTP * SSLD + 2 * H + 2 * OUTP + 4 * (size_t)H * SSLD +
                      2 * (size_t)OUTP * SSLD + 64 + (size_t)H * SSLD + (size_t)H * (OUTP + 4) + 3 * OUTP;
  const size_t help = (size_t)H * SXLD + 16 * SKQ * (size_t)SSLD + 16 + 2 * (size_t)nmb;
  return (lead > help ? lead : help) * sizeof(float);
}

extern "C" bool osa_is_exchange_ptr(const void* p);  // ppo_pass_kernel.hip

template <int OT, bool DP = false>
static int osa_launch_split(const OsaSplitArgs& a, hipStream_t stream) {
  static OsaPerDeviceOnce attr_set;
  const size_t lds = osa_split_lds_bytes(OT, a.nmb);
  if (lds > 160 * 1024) return OSA_EUNSUPPORTED;
  if (attr_set.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&osa_wide_split_kernel<OT, DP>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return OSA_EHIP;
    attr_set.set();
  }
  // the workgroups of a network wait for each other every step, so they MUST be co-resident: a cooperative
  // launch makes the runtime verify that and refuses otherwise (the caller then takes the one-CU kernel)
  OsaSplitArgs arg = a;
  void* kargs[] = {&arg};
  const int nblk = DP ? (a.dp_place == 1 ? 8 * ((3 * (a.C + 1) + 7) / 8) * a.world : a.world * 3 * (a.C + 1))
                      : (a.local == 1 ? 8 * (a.C + 1) : 3 * (a.C + 1));
  const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&osa_wide_split_kernel<OT, DP>),
                                                  dim3(nblk), dim3(256), kargs,
                                                  (unsigned)lds, stream);
  if (e == hipSuccess) return OSA_OK;
  (void)hipGetLastError();
  return e == hipErrorCooperativeLaunchTooLarge ? OSA_EUNSUPPORTED : OSA_EHIP;
}

extern "C" {

int osa_ppo_split_pass_supported(int obs_dim, int act_dim, int hidden) {
  if (hidden != 64 || obs_dim < 65 || obs_dim > 16 * SKQ * SCMAX || obs_dim > 512 || act_dim < 1 || act_dim > 32)
    return 0;
  return 1;
}

size_t osa_ppo_split_pass_xch_floats(int obs_dim, int act_dim, int hidden) {
  if (!osa_ppo_split_pass_supported(obs_dim, act_dim, hidden)) return 0;
  return (size_t)128 + 3 * (size_t)SXNET;
}

int osa_ppo_split_pass(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                       int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                       const float* logp, const float* target_value_r, const float* target_value_c,
                       const float* adv_r, const float* adv_c, const long* perm, long M, int B,
                       const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                       float* xch, int local, float* step_stats, void* stream) {
  if (!osa_ppo_split_pass_supported(obs_dim, act_dim, hidden)) return OSA_EUNSUPPORTED;
  if (B > 64 || loss_kind < 0 || loss_kind > 1) return OSA_EUNSUPPORTED;  // larger batches: per-step kernels
  if ((M + B - 1) / B > 8192) return OSA_EUNSUPPORTED;  // the helpers tabulate Adam's bias corrections in LDS
  OSA_REQUIRE(params && adam_m && adam_v && adam_step && obs && act && logp && hp && step_stats && xch);
  OSA_REQUIRE(target_value_r && target_value_c && adv_r && adv_c && M > 0 && B > 0);
  OSA_REQUIRE(ld_obs >= obs_dim && ld_act >= act_dim);
  if (!local && !osa_is_exchange_ptr(xch)) return OSA_EINVAL;  // hand-offs across XCCs rely on uncached memory
  if (local && osa_is_exchange_ptr(xch)) return OSA_EINVAL;
  if (ld_obs % 4 != 0 || (reinterpret_cast<uintptr_t>(obs) & 15) != 0) return OSA_EUNSUPPORTED;  // pad the rows
  if ((double)M * ld_obs >= 2147483647.0 * 4) return OSA_EUNSUPPORTED;
  OsaSplitArgs a = {};
  a.xch = xch;
  a.local = local;  // 0, 1, or 3 (see the kernel)
  a.nd = osa_make_net(obs_dim, act_dim, hidden);
  a.C = (a.nd.KB + SKQ - 1) / SKQ;
  a.params = params; a.adam_m = adam_m; a.adam_v = adam_v; a.adam_step = adam_step;
  a.obs = obs; a.ld_obs = ld_obs; a.act = act; a.ld_act = ld_act; a.logp = logp;
  a.tgt_r = target_value_r; a.tgt_c = target_value_c; a.adv_r = adv_r; a.adv_c = adv_c;
  a.perm = perm; a.M = M; a.B = B; a.nmb = (int)((M + B - 1) / B); a.lagrange = lagrange;
  a.hp.clip = hp->clip; a.hp.entropy_coef = hp->entropy_coef;
  a.hp.critic_norm_coef = hp->critic_norm_coef; a.hp.max_grad_norm = hp->max_grad_norm;
  a.hp.lr_actor = hp->lr_actor; a.hp.lr_critic = hp->lr_critic; a.hp.beta1 = hp->beta1;
  a.hp.beta2 = hp->beta2; a.hp.adam_eps = hp->adam_eps; a.hp.use_critic_norm = hp->use_critic_norm;
  a.hp.use_max_grad_norm = hp->use_max_grad_norm; a.hp.use_cost = hp->use_cost;
  a.loss_kind = loss_kind; a.nets_mask = nets_mask & (hp->use_cost ? 7 : 3); a.stats = step_stats;
  hipStream_t st = osa_stream(stream);
  // flag words are step counters of THIS launch (the sticky error word, SF_ERR, survives)
  if (hipMemsetAsync(xch, 0, SF_ERR * sizeof(int), st) != hipSuccess) return OSA_EHIP;
  if (hipMemsetAsync(reinterpret_cast<int*>(xch) + SF_ARRIVE, 0, sizeof(int), st) != hipSuccess) return OSA_EHIP;
  // data regions: all-ones = "not published" for the dz1 buffers (SX_SENT) and step -1 for the squared-norm slots
  // (which carry their own step counters)
  if (hipMemsetAsync(xch + 128, 0xFF, 3 * (size_t)SXNET * sizeof(float), st) != hipSuccess) return OSA_EHIP;
  const int OT = a.nd.OUTP / 16;
  if (OT == 1) return osa_launch_split<1>(a, st);
  if (OT == 2) return osa_launch_split<2>(a, st);
  return OSA_EUNSUPPORTED;
}

size_t osa_ppo_split_dp_dpx_floats(int obs_dim, int act_dim, int hidden, int world) {
  if (!osa_ppo_split_pass_supported(obs_dim, act_dim, hidden) || world < 1) return 0;
  return (size_t)SDPH + (size_t)2 * 3 * (SCMAX + 1) * world * SDPW;
}

size_t osa_ppo_split_dp_xch_floats(int obs_dim, int act_dim, int hidden, int world) {
  if (!osa_ppo_split_pass_supported(obs_dim, act_dim, hidden) || world < 1) return 0;
  const size_t per_rank = (size_t)128 + 3 * (size_t)SXNET;
  return (size_t)world * per_rank + osa_ppo_split_dp_dpx_floats(obs_dim, act_dim, hidden, world);
}

int osa_ppo_split_dp_pass(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                          int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                          const float* logp, const float* target_value_r, const float* target_value_c,
                          const float* adv_r, const float* adv_c, const long* perm, long M, int B, int world,
                          const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                          float* xch, float* dpx, int place, float* step_stats, void* stream) {
  if (!osa_ppo_split_pass_supported(obs_dim, act_dim, hidden)) return OSA_EUNSUPPORTED;
  if (B > 64 || loss_kind < 0 || loss_kind > 1) return OSA_EUNSUPPORTED;
  OSA_REQUIRE((place == 1 || place == 3) ? dpx != nullptr : dpx == nullptr);
  if (dpx && osa_is_exchange_ptr(dpx)) return OSA_EINVAL;  // the owner group's L2 serves ordinary memory
  if ((M + B - 1) / B > 8192) return OSA_EUNSUPPORTED;  // the helpers tabulate Adam's bias corrections in LDS
  OSA_REQUIRE(params && adam_m && adam_v && adam_step && obs && act && logp && hp && step_stats && xch);
  OSA_REQUIRE(target_value_r && target_value_c && adv_r && adv_c && M > 0 && B > 0 && world >= 1);
  OSA_REQUIRE(ld_obs >= obs_dim && ld_act >= act_dim);
  if (!osa_is_exchange_ptr(xch)) return OSA_EINVAL;  // hand-offs across XCCs rely on uncached memory
  if (ld_obs % 4 != 0 || (reinterpret_cast<uintptr_t>(obs) & 15) != 0) return OSA_EUNSUPPORTED;  // pad the rows
  if ((double)M * world * ld_obs >= 2147483647.0 * 4) return OSA_EUNSUPPORTED;
  OsaSplitArgs a = {};
  a.nd = osa_make_net(obs_dim, act_dim, hidden);
  a.C = (a.nd.KB + SKQ - 1) / SKQ;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return OSA_EHIP;
  if (world * 3 * (a.C + 1) > cus) return OSA_EUNSUPPORTED;  // one workgroup per CU, all co-resident
  // (place: ceil(groups / 8) x world workgroups per XCC; 3: test hook -- the placed protocol on the rank-major grid)
  if (place == 1 && ((3 * (a.C + 1) + 7) / 8) * world > cus / 8) return OSA_EUNSUPPORTED;
  const size_t per_rank = (size_t)128 + 3 * (size_t)SXNET;
  a.xch = xch;
  a.local = 0;
  a.world = world;
  a.rank_xch = (long)per_rank;
  a.dpx = dpx ? dpx : xch + (size_t)world * per_rank;
  a.dp_hdr = reinterpret_cast<int*>(xch + (size_t)world * per_rank);
  a.dp_place = place;
  a.params = params; a.adam_m = adam_m; a.adam_v = adam_v; a.adam_step = adam_step;
  a.obs = obs; a.ld_obs = ld_obs; a.act = act; a.ld_act = ld_act; a.logp = logp;
  a.tgt_r = target_value_r; a.tgt_c = target_value_c; a.adv_r = adv_r; a.adv_c = adv_c;
  a.perm = perm; a.M = M; a.B = B; a.nmb = (int)((M + B - 1) / B); a.lagrange = lagrange;
  a.hp.clip = hp->clip; a.hp.entropy_coef = hp->entropy_coef;
  a.hp.critic_norm_coef = hp->critic_norm_coef; a.hp.max_grad_norm = hp->max_grad_norm;
  a.hp.lr_actor = hp->lr_actor; a.hp.lr_critic = hp->lr_critic; a.hp.beta1 = hp->beta1;
  a.hp.beta2 = hp->beta2; a.hp.adam_eps = hp->adam_eps; a.hp.use_critic_norm = hp->use_critic_norm;
  a.hp.use_max_grad_norm = hp->use_max_grad_norm; a.hp.use_cost = hp->use_cost;
  a.loss_kind = loss_kind; a.nets_mask = nets_mask & (hp->use_cost ? 7 : 3); a.stats = step_stats;
  hipStream_t st = osa_stream(stream);
  // flag words and arrival counters are step counters of THIS launch (rank 0's sticky error word survives)
  for (int r = 0; r < world; ++r) {
    float* xr = xch + (size_t)r * per_rank;
    if (hipMemsetAsync(xr, 0, SF_ERR * sizeof(int), st) != hipSuccess) return OSA_EHIP;
    if (hipMemsetAsync(xr + 128, 0xFF, 3 * (size_t)SXNET * sizeof(float), st) != hipSuccess) return OSA_EHIP;
  }
  if (hipMemsetAsync(a.dp_hdr, 0, SDPH * sizeof(int), st) != hipSuccess) return OSA_EHIP;
  const int OT = a.nd.OUTP / 16;
  if (OT == 1) return osa_launch_split<1, true>(a, st);
  if (OT == 2) return osa_launch_split<2, true>(a, st);
  return OSA_EUNSUPPORTED;
}

// != 0 when a workgroup of any split pass since the allocation of `xch` gave up waiting for a peer (1) or the
// workgroups of a network of a `local` pass were NOT placed on one XCC (2): results invalid
int osa_ppo_split_pass_timed_out(const float* xch, int* out) {
  OSA_REQUIRE(xch && out);
  return hipMemcpy(out, reinterpret_cast<const int*>(xch) + SF_ERR, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess
             ? OSA_OK : OSA_EHIP;
}

// clears the sticky word (after a tripped PLACEMENT check -- flag 2 -- nothing was modified and the caller repeats the
// pass with another placement; a time-out -- flag 1 -- must not be cleared: the replica is no longer trustworthy)
int osa_ppo_split_pass_clear_flag(float* xch) {
  OSA_REQUIRE(xch);
  return hipMemset(reinterpret_cast<int*>(xch) + SF_ERR, 0, sizeof(int)) == hipSuccess ? OSA_OK : OSA_EHIP;
}

}  // extern "C"
