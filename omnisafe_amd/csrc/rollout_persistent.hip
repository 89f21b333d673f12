// The device part of an epoch's rollout as ONE persistent launch (round 4).
//
// OnPolicyAdapter.rollout (omnisafe/adapter/onpolicy_adapter.py:58-136) on a device-resident env is, per vector
// step, five small launches -- policy step, env step, normaliser push, normalise, episode accounting: 38 us of
// launch latency for 0.1-2 MB of data each (0.61 ms per 16-step epoch of the large-batch benchmark).
// A vector step has ONE global dependency: the normaliser's batch statistics over ALL envs of the new observations
// (Normalizer._push, omnisafe/common/normalizer.py:109-139).  Everything else is local to an env.  Here a workgroup
// owns 128 envs for the whole epoch (8 waves x 16 rows for the three networks' forward passes) and the workgroups
// meet at a software grid barrier once per normaliser push:
//
//   prologue  env reset -> raw observations (LDS) -> partial moments -> BARRIER -> merge -> normalise -> buffer row 0
//   step t    policy step on row t (osa_policy_rows: the very function of osa_policy_step_kernel) -> act, values,
//             logp rows t; env step (Philox) -> raw next / final observations (LDS), reward, cost, truncation;
//             [truncation: masked push of the final observations -> BARRIER -> merge -> normalise -> V(final)]
//             push of the next observations -> BARRIER -> merge -> normalise -> buffer row t + 1
//             [epoch end: V(next)];  episode accounting + bootstrap selection (osa_rollout_post_step_kernel's rows)
//
// Same arithmetic as the separate kernels, bit for bit: the partial moments are the SAME 128-row x 64-column blocks
// reduced by the same thread pattern (rollout_kernels.hip osa_norm_push_kernel: block index = workgroup index here),
// published as agent-scope 8-byte words and merged in block order -- by EVERY workgroup (same order, same bits; each
// keeps the running state in LDS, workgroup 0 writes it back at the end); the Philox counters of the env and of the
// policy noise advance exactly as the host-side counters of the launch-per-step path do.
// tests/test_rollout_gpu.py::test_persistent_rollout_equals_the_launch_per_step_rollout compares every buffer row,
// the normaliser state, the episode rows and the stream positions of both paths with torch.equal.
//
// Applies to: SynthVectorEnv (the benchmark's env: obs ~ N(0,1), truncation every `horizon` steps for all envs at
// once), plain OnPolicyAdapter (no Saute / Simmer hook, no reward / cost normaliser), fused [64, 64] tanh networks,
// N a multiple of 128, obs_dim <= 96.  Everything else keeps the launches (adapter.py).
#include <stdlib.h>

#include "mlp_device.h"
#include "policy_rows.h"

#define ORP_ROWS 128  // envs per workgroup = rows per partial-moment block (OSA_NORM_ROWS of rollout_kernels.hip)
#define ORP_STAGE_BLOCKS 32  // blocks' partial sums staged through LDS per trip of the merge

struct OsaRollArgs {
  OsaNet nd;
  const float* params;
  int N, T, D, A;
  // rollout buffer, time-major
  float* obs;
  float* act;
  float* value_r;
  float* value_c;
  float* logp;
  float* reward;
  float* cost;
  uint8_t* path_end;
  float* boot_r;
  float* boot_c;
  // episode rows / state
  uint8_t* ep_done;
  float* ep_ret_out;
  float* ep_cost_out;
  float* ep_len_out;
  float* ep_ret;
  float* ep_cost;
  float* ep_len;
  float* last_obs;    // [N][D] normalised observation after the last step
  float* final_norm;  // [N][D] normalised final observations of the last truncation step
  float* act_env;     // [N][A] scaled action of the last step
  const float* old_min;
  const float* old_max;
  float* vscratch;    // [4][N]: V(final) r / c, V(next) r / c
  // normaliser
  float* n_mean;
  float* n_sumsq;
  float* n_var;
  float* n_std;
  long* n_count;
  float clip;
  double* ws;          // [2][nrb D 2 + nrb] partial sums, double-buffered by barrier parity
  unsigned int* bar;   // [0] arrival counter (zeroed per launch), [1] sticky time-out flag
  // env (SynthVectorEnv)
  unsigned long long env_seed, env_step0;
  const unsigned long long* env_step_base;
  int horizon;
  float cost_p;
  int* steps;
  float* env_obs;      // [N][D] raw observation of the last step (what env.step returned last)
  float* env_final;    // [N][D]
  float* env_reward;
  float* env_cost;
  uint8_t* env_term;
  uint8_t* env_trunc;
  // policy noise
  unsigned long long pi_seed, pi_off0;
  const unsigned long long* pi_off_base;
  int actor_in_lds;   // the actor's parameter block is copied into LDS once (padded rows) and read from there
  int defer_critics;  // V_r / V_c of the buffer rows are evaluated by the caller afterwards (one launch over T N rows)
  long long* dbg;  // optional phase clocks of workgroup 0 (osa_debug_set_rollout_clock_buffer), 100 MHz ticks
};

// phase clocks (workgroup 0, thread 0): 0 policy step, 1 env step, 2 partial moments, 3 grid barrier, 4 merge,
// 5 normalise, 6 bootstrap values, 7 episode accounting
struct OrpClk {
  bool on;
  long long t;
  long long acc[8];
};
#define ORP_MARK(clk, k)                    \
  do {                                      \
    if ((clk).on) {                         \
      const long long now_ = wall_clock64(); \
      (clk).acc[k] += now_ - (clk).t;       \
      (clk).t = now_;                       \
    }                                       \
  } while (0)

__device__ __forceinline__ void orp_ws_put(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double orp_ws_get(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p),
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// all workgroups have published (agent-scope stores, waited for) what the others are about to read
__device__ __forceinline__ void orp_grid_barrier(unsigned int* bar, unsigned int target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int seen = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    int spins = 0;
    while (seen < target) {
      __builtin_amdgcn_s_sleep(1);
      seen = __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (++spins > (1 << 22)) {  // never hang the device: flag it (the adapter checks at its next synchronisation)
        __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
}

struct OrpShared {
  double s1[4][64], s2[4][64];
  int scnt[4];
  long count;
  long n_raw;
  float mean[96], sumsq[96], var[96], stdv[96];
  int steps[ORP_ROWS];
  float rew[ORP_ROWS], cst[ORP_ROWS];
  uint8_t trunc[ORP_ROWS];
};

// Normalizer._push over `x` (this workgroup's 128 raw rows in LDS, leading dimension D) with an optional row mask:
// the partial sums of osa_norm_push_kernel for block `blockIdx.x`, published into `ws`; then the grid barrier; then
// the merge of ALL blocks' partials in block order into this workgroup's LDS copy of the running state.
__device__ void orp_push(const OsaRollArgs& a, OrpShared& sh, const float* __restrict__ x, const uint8_t* __restrict__ mask,
                         double* __restrict__ ws, unsigned int barrier_target, double* __restrict__ stg, OrpClk& clk) {
#pragma clang fp contract(off)
  const int D = a.D, nrb = gridDim.x;
  const int tid = threadIdx.x;
  const int ncb = (D + 63) / 64;
  for (int cb = 0; cb < ncb; ++cb) {
    if (tid < 256) {
      const int cx = tid & 63, ry = tid >> 6;
      const int col = cb * 64 + cx;
      const double c = (col < D) ? (double)sh.mean[col] : 0.0;
      double a1 = 0.0, a2 = 0.0;
      int cnt = 0;
      const int cl = (col < D) ? col : 0;
      for (int rb = ry; rb < ORP_ROWS; rb += 32) {
        float xv[8];
        uint8_t mv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int r = min(rb + 4 * u, ORP_ROWS - 1);
          xv[u] = x[r * D + cl];
          mv[u] = mask ? mask[r] : (uint8_t)1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (rb + 4 * u < ORP_ROWS && mv[u] != 0) {
            ++cnt;
            if (col < D) {
              const double d = (double)xv[u] - c;
              a1 += d;
              a2 = __builtin_fma(d, d, a2);
            }
          }
        }
      }
      sh.s1[ry][cx] = a1;
      sh.s2[ry][cx] = a2;
      if (cx == 0) sh.scnt[ry] = cnt;
    }
    __syncthreads();
    if (tid < 64) {
      const int cx = tid, col = cb * 64 + cx;
      if (col < D) {
        double* o = ws + ((long)blockIdx.x * D + col) * 2;
        orp_ws_put(o, sh.s1[0][cx] + sh.s1[1][cx] + sh.s1[2][cx] + sh.s1[3][cx]);
        orp_ws_put(o + 1, sh.s2[0][cx] + sh.s2[1][cx] + sh.s2[2][cx] + sh.s2[3][cx]);
      }
      if (cx == 0 && cb == 0)
        orp_ws_put(ws + (long)nrb * D * 2 + blockIdx.x, (double)(sh.scnt[0] + sh.scnt[1] + sh.scnt[2] + sh.scnt[3]));
    }
    __syncthreads();
  }
  ORP_MARK(clk, 2);
  orp_grid_barrier(a.bar, barrier_target);
  ORP_MARK(clk, 3);
  // ---- merge (every workgroup: same order, same bits).  The row count is a sum of small integers (exact in any
  // order): one wave adds it up; the partial sums are staged through LDS by ALL threads (independent loads in flight)
  // and then added per column in block order, as osa_norm_push_kernel's last workgroup does
  if (tid < 64) {
    double n = 0.0;
    for (int b = tid; b < nrb; b += 64) n += orp_ws_get(ws + (long)nrb * D * 2 + b);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) n += __shfl_xor(n, o);
    if (tid == 0) sh.n_raw = (long)n;
  }
  __syncthreads();
  const long n_raw = sh.n_raw;
  if (n_raw == 0) return;  // (block-uniform)
  const long cnt_old = sh.count;
  const long cnt_new = cnt_old + n_raw;
  double S1 = 0.0, S2 = 0.0;
  const int c0 = tid;  // one column per thread (D <= 96)
  for (int b0 = 0; b0 < nrb; b0 += ORP_STAGE_BLOCKS) {
    const int nb = min(ORP_STAGE_BLOCKS, nrb - b0);
    const double* src = ws + (long)b0 * D * 2;
#pragma unroll 4
    for (int e = tid; e < nb * D * 2; e += blockDim.x) stg[e] = orp_ws_get(src + e);
    __syncthreads();
    if (c0 < D) {
      for (int b = 0; b < nb; ++b) {
        S1 += stg[(b * D + c0) * 2];
        S2 += stg[(b * D + c0) * 2 + 1];
      }
    }
    __syncthreads();
  }
  if (c0 < D) {
    const float mean_old = sh.mean[c0];
    const double c = (double)mean_old;
    const float mean_raw = (float)(c + S1 / (double)n_raw);
    double q = S2 - S1 * S1 / (double)n_raw;
    if (q < 0.0) q = 0.0;
    const float sumq_raw = (float)q;
    float m_new, ss_new;
    if (cnt_old == 0) {
      m_new = mean_raw;
      ss_new = sumq_raw;
    } else {
      const float delta = mean_raw - mean_old;
      m_new = mean_old + delta * (float)n_raw / (float)cnt_new;
      ss_new = sh.sumsq[c0] + (sumq_raw + delta * delta * (float)cnt_old * (float)n_raw / (float)cnt_new);
    }
    sh.mean[c0] = m_new;
    sh.sumsq[c0] = ss_new;
    const float v = ss_new / (float)(cnt_new - 1);  // count == 1 -> 0/0 = NaN, as the reference
    sh.var[c0] = v;
    const float s = sqrtf(v);
    float sd = fmaxf(s, 1e-2f);
    if (s != s) sd = s;
    sh.stdv[c0] = sd;
  }
  __syncthreads();
  if (tid == 0) sh.count = cnt_new;
  __syncthreads();
  ORP_MARK(clk, 4);
}

// osa_normalize_kernel for this workgroup's rows: y[row][col] (global) = clamp((x - mean) / std) where the row is
// selected (mask) and the running count exceeds 1, else x
__device__ void orp_normalize(const OsaRollArgs& a, const OrpShared& sh, const float* __restrict__ x,
                              const uint8_t* __restrict__ mask, float* __restrict__ y_rows) {
#pragma clang fp contract(off)
  const int D = a.D;
  const bool have = sh.count > 1;
  // element e = r D + col of the workgroup's contiguous 128 x D block; (r, col) advance without integer divisions
  const int dq = (int)blockDim.x / D, dr = (int)blockDim.x - dq * D;
  int r = (int)threadIdx.x / D, col = (int)threadIdx.x - r * D;
  for (int e = threadIdx.x; e < ORP_ROWS * D; e += blockDim.x) {
    float v = x[e];
    const bool on = mask == nullptr || mask[r] != 0;
    if (on && have) {
      v = (v - sh.mean[col]) / sh.stdv[col];
      v = fminf(fmaxf(v, -a.clip), a.clip);
    }
    y_rows[e] = v;
    r += dq;
    col += dr;
    if (col >= D) {
      col -= D;
      ++r;
    }
  }
}

// osa_synth_env_kernel for this workgroup's rows at stream position `step`: raw observations into xs (LDS), final
// observations into fs where the env truncates, reward / cost / truncation / step counters into LDS
__device__ void orp_env(const OsaRollArgs& a, OrpShared& sh, unsigned long long step, int r0, float* __restrict__ xs,
                        float* __restrict__ fs, bool reset_only) {
  const int D = a.D, npair = (D + 1) / 2;
  const int dq = (int)blockDim.x / npair, dr = (int)blockDim.x - dq * npair;
  int r = (int)threadIdx.x / npair, pair = (int)threadIdx.x - r * npair;
  for (int e = threadIdx.x; e < ORP_ROWS * npair; e += blockDim.x) {
    const int n = r0 + r;
    uint8_t trunc = 0;
    if (!reset_only) trunc = (sh.steps[r] + 1 >= a.horizon) ? 1 : 0;
    uint32_t w[4];
    osa_philox(a.env_seed, step, ((unsigned long long)n << 20) + pair, w);
    float p, q;
    osa_box_muller(w[0], w[1], p, q);
    const int i0 = 2 * pair, i1 = 2 * pair + 1;
    if (trunc) {  // (the second pair of normals -- the post-reset observation -- is only drawn where it is used)
      float c2, d2;
      osa_box_muller(w[2], w[3], c2, d2);
      fs[r * D + i0] = p;
      if (i1 < D) fs[r * D + i1] = q;
      xs[r * D + i0] = c2;
      if (i1 < D) xs[r * D + i1] = d2;
    } else {
      xs[r * D + i0] = p;
      if (i1 < D) xs[r * D + i1] = q;
    }
    r += dq;
    pair += dr;
    if (pair >= npair) {
      pair -= npair;
      ++r;
    }
  }
  __syncthreads();  // every pair of a row has read steps[] before it changes
  if (threadIdx.x < ORP_ROWS) {
    const int r = threadIdx.x, n = r0 + r;
    if (reset_only) {
      sh.steps[r] = 0;
    } else {
      const uint8_t trunc = (sh.steps[r] + 1 >= a.horizon) ? 1 : 0;
      uint32_t w[4];
      osa_philox(a.env_seed ^ 0x9E3779B97F4A7C15ull, step, (unsigned long long)n, w);
      float p, q;
      osa_box_muller(w[0], w[1], p, q);
      sh.rew[r] = p;
      sh.cst[r] = (osa_u01(w[2]) <= a.cost_p) ? 1.f : 0.f;
      sh.trunc[r] = trunc;
      sh.steps[r] = trunc ? 0 : sh.steps[r] + 1;
    }
  }
  __syncthreads();
}

template <int HT, int OT>
__global__ __launch_bounds__(512) void osa_rollout_persistent_kernel(OsaRollArgs a) {
  extern __shared__ __attribute__((aligned(16))) float orp_smem[];
  __shared__ OrpShared sh;
  const int D = a.D, N = a.N, T = a.T, A = a.A;
  float* xs = orp_smem;               // [128][D] raw next observations
  float* fs = xs + ORP_ROWS * D;      // [128][D] raw final observations
  double* stg = reinterpret_cast<double*>(fs + ORP_ROWS * D);  // [ORP_STAGE_BLOCKS][D][2] staged partial sums
  OrpClk clk;
  clk.on = a.dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
  clk.t = clk.on ? wall_clock64() : 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) clk.acc[i] = 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15;
  const int r0 = blockIdx.x * ORP_ROWS;
  const int nrb = gridDim.x;
  const long wsz = (long)nrb * D * 2 + nrb;
  const unsigned long long env_base = a.env_step0 + (a.env_step_base ? *a.env_step_base : 0ull);
  const unsigned long long pi_base = a.pi_off0 + (a.pi_off_base ? *a.pi_off_base : 0ull);
  // ---- running normaliser state -> LDS
  for (int c = tid; c < D; c += blockDim.x) {
    sh.mean[c] = a.n_mean[c];
    sh.sumsq[c] = a.n_sumsq[c];
    sh.var[c] = a.n_var[c];
    sh.stdv[c] = a.n_std[c];
  }
  if (tid == 0) sh.count = *a.n_count;
  if (tid < ORP_ROWS) sh.steps[tid] = 0;
  __syncthreads();
  // ---- the actor's parameters -> LDS, rows padded by 4 floats (16 lanes x 16-byte fragments at a 272-byte stride hit
  // 16 distinct bank groups): the per-step forward pass of the 16 rows of a wave otherwise waits on L2 for every
  // weight fragment (same values, same MFMA order: same bits)
  OsaNet nl = a.nd;
  const float* pi_params = a.params;
  if (a.actor_in_lds) {
    float* wl = reinterpret_cast<float*>(stg + ORP_STAGE_BLOCKS * D * 2);
    const int H = a.nd.H, INP = a.nd.INP, OUTP = a.nd.OUTP;
    nl.INP = INP + 4;
    nl.H = H + 4;
    nl.oW1 = 0;
    nl.ob1 = nl.oW1 + H * nl.INP;
    nl.oW2 = nl.ob1 + H;
    nl.ob2 = nl.oW2 + H * nl.H;
    nl.oW3 = nl.ob2 + H;
    nl.ob3 = nl.oW3 + OUTP * nl.H;
    nl.oLS = nl.ob3 + OUTP;
    nl.P = nl.oLS + OUTP;
    for (int e = tid; e < H * INP; e += blockDim.x) {
      const int r = e / INP, c = e - r * INP;
      wl[nl.oW1 + r * nl.INP + c] = a.params[a.nd.oW1 + e];
    }
    for (int e = tid; e < H * H; e += blockDim.x) {
      const int r = e / H, c = e - r * H;
      wl[nl.oW2 + r * nl.H + c] = a.params[a.nd.oW2 + e];
    }
    for (int e = tid; e < OUTP * H; e += blockDim.x) {
      const int r = e / H, c = e - r * H;
      wl[nl.oW3 + r * nl.H + c] = a.params[a.nd.oW3 + e];
    }
    for (int e = tid; e < H; e += blockDim.x) {
      wl[nl.ob1 + e] = a.params[a.nd.ob1 + e];
      wl[nl.ob2 + e] = a.params[a.nd.ob2 + e];
    }
    for (int e = tid; e < OUTP; e += blockDim.x) {
      wl[nl.ob3 + e] = a.params[a.nd.ob3 + e];
      wl[nl.oLS + e] = a.params[a.nd.oLS + e];
    }
    pi_params = wl;
  }
  __syncthreads();
  unsigned int nbar = 0;  // barriers passed so far
  // ---- prologue: reset, first push, row 0 (onpolicy_adapter.py:80-84)
  orp_env(a, sh, env_base, r0, xs, fs, true);
  ORP_MARK(clk, 1);
  orp_push(a, sh, xs, nullptr, a.ws + (long)(nbar & 1) * wsz, (nbar + 1) * nrb, stg, clk);
  ++nbar;
  orp_normalize(a, sh, xs, nullptr, a.obs + (long)r0 * D);
  __syncthreads();  // the rows just written are read back by this workgroup's policy step
  ORP_MARK(clk, 5);
  const bool vec_ok = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.obs) & 15) == 0);
  const int row = r0 + 16 * wave + j;  // this lane's env in the forward passes
  float er = 0.f, ec = 0.f, el = 0.f;  // this thread's env (tid < 128): running episode sums
  for (int t = 0; t < T; ++t) {
    const long tN = (long)t * N;
    const bool epoch_end = t >= T - 1;
    // ---- policy step on row t: agent.step's k-th call of the epoch, k = t + 1 + (value evaluations so far)
    const unsigned long long k = (unsigned long long)(t + 1 + t / a.horizon);
    const float* xrow = a.obs + (tN + row) * D;
    osa_policy_rows<HT, OT>(nl, pi_params, 0, xrow, D, vec_ok, row, true, nullptr, a.pi_seed, pi_base + k, 0, a.act + tN * A, A,
                            a.value_r + tN, a.value_c + tN, a.logp + tN, nullptr, 0, a.act_env, A, a.old_min, a.old_max,
                            -1.f, 1.f);
    if (!a.defer_critics) {
#pragma unroll 1
      for (int net = 1; net < 3; ++net)
        osa_policy_rows<HT, OT>(a.nd, a.params, net, xrow, D, vec_ok, row, true, nullptr, a.pi_seed, pi_base + k, 0,
                                a.act + tN * A, A, a.value_r + tN, a.value_c + tN, a.logp + tN, nullptr, 0, a.act_env, A,
                                a.old_min, a.old_max, -1.f, 1.f);
    }
    ORP_MARK(clk, 0);
    // ---- env step (SynthVectorEnv ignores the action)
    orp_env(a, sh, env_base + 1 + t, r0, xs, fs, false);
    ORP_MARK(clk, 1);
    const bool have_final = ((t + 1) % a.horizon) == 0;  // every env truncates on this step (they reset together)
    float* vfr = a.vscratch;
    float* vfc = a.vscratch + N;
    float* vnr = a.vscratch + 2 * (long)N;
    float* vnc = a.vscratch + 3 * (long)N;
    if (have_final) {  // ObsNormalize.step on info['final_observation'] (envs/wrapper.py:231-241), then V(final)
      orp_push(a, sh, fs, sh.trunc, a.ws + (long)(nbar & 1) * wsz, (nbar + 1) * nrb, stg, clk);
      ++nbar;
      orp_normalize(a, sh, fs, sh.trunc, a.final_norm + (long)r0 * D);
      ORP_MARK(clk, 5);
    }
    orp_push(a, sh, xs, nullptr, a.ws + (long)(nbar & 1) * wsz, (nbar + 1) * nrb, stg, clk);
    ++nbar;
    float* nxt = epoch_end ? a.last_obs : a.obs + (tN + N) * D;
    orp_normalize(a, sh, xs, nullptr, nxt + (long)r0 * D);
    __syncthreads();
    ORP_MARK(clk, 5);
    if (have_final) {
      const float* frow = a.final_norm + (long)row * D;
#pragma unroll 1
      for (int net = 1; net < 3; ++net)
        osa_policy_rows<HT, OT>(a.nd, a.params, net, frow, D, vec_ok, row, true, nullptr, 0ull, 0ull, 1, nullptr, 0, vfr,
                                vfc, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, -1.f, 1.f);
    }
    if (epoch_end) {
      const float* nrow = a.last_obs + (long)row * D;
#pragma unroll 1
      for (int net = 1; net < 3; ++net)
        osa_policy_rows<HT, OT>(a.nd, a.params, net, nrow, D, vec_ok, row, true, nullptr, 0ull, 0ull, 1, nullptr, 0, vnr,
                                vnc, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, -1.f, 1.f);
    }
    __syncthreads();  // the values written above are read by other lanes below
    ORP_MARK(clk, 6);
    // ---- episode accounting + bootstrap selection (osa_rollout_post_step_kernel), one thread per env; the running
    // sums start at zero every epoch (_reset_log, onpolicy_adapter.py:78) and stay in registers
    if (tid < ORP_ROWS) {
      const int r = tid, n = r0 + r;
      const float rw = sh.rew[r], cs = sh.cst[r];
      a.reward[tN + n] = rw;
      a.cost[tN + n] = cs;
      er += rw;
      ec += cs;
      el += 1.f;
      const bool done = false, time_out = sh.trunc[r] != 0;
      uint8_t pe = 0, ed = 0;
      float lr = 0.f, lc = 0.f;
      if (epoch_end || done || time_out) {
        if (!done) {
          if (epoch_end) {
            lr = vnr[n];
            lc = vnc[n];
          }
          if (time_out && have_final) {
            lr = vfr[n];
            lc = vfc[n];
          }
        }
        pe = 1;
        if (done || time_out) {
          ed = 1;
          a.ep_ret_out[tN + n] = er;
          a.ep_cost_out[tN + n] = ec;
          a.ep_len_out[tN + n] = el;
          er = ec = el = 0.f;
        }
      }
      a.path_end[tN + n] = pe;
      a.boot_r[tN + n] = lr;
      a.boot_c[tN + n] = lc;
      a.ep_done[tN + n] = ed;
    }
    __syncthreads();
    ORP_MARK(clk, 7);
  }
  if (clk.on) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a.dbg[i] = clk.acc[i];
  }
  // ---- what the launch-per-step path leaves behind: the env's last outputs and step counters, the running state
  for (int e = tid; e < ORP_ROWS * D; e += blockDim.x) {
    a.env_obs[(long)r0 * D + e] = xs[e];
    if (T >= a.horizon) a.env_final[(long)r0 * D + e] = fs[e];
  }
  if (tid < ORP_ROWS) {
    const int n = r0 + tid;
    a.steps[n] = sh.steps[tid];
    a.ep_ret[n] = er;
    a.ep_cost[n] = ec;
    a.ep_len[n] = el;
    a.env_reward[n] = sh.rew[tid];
    a.env_cost[n] = sh.cst[tid];
    a.env_term[n] = 0;
    a.env_trunc[n] = sh.trunc[tid];
  }
  if (blockIdx.x == 0) {
    for (int c = tid; c < D; c += blockDim.x) {
      a.n_mean[c] = sh.mean[c];
      a.n_sumsq[c] = sh.sumsq[c];
      a.n_var[c] = sh.var[c];
      a.n_std[c] = sh.stdv[c];
    }
    if (tid == 0) *a.n_count = sh.count;
  }
}

static long long* g_orp_clocks = nullptr;

static bool osa_env_is_zero(const char* name) {
  const char* v = getenv(name);
  return v != nullptr && v[0] == '0' && v[1] == 0;
}

extern "C" {

// phase clocks of workgroup 0 (8 x int64 of device memory, 100 MHz ticks summed over the epoch); NULL switches them off
int osa_debug_set_rollout_clock_buffer(long long* dev_ptr) {
  g_orp_clocks = dev_ptr;
  return OSA_OK;
}

int osa_rollout_persistent_supported(int obs_dim, int act_dim, int hidden, int N) {
  if ((hidden & 0xFFFF) != 64 || ((hidden >> 16) & 0xF) != OSA_ACT_TANH) return 0;
  if (obs_dim < 1 || obs_dim > 96 || act_dim < 1 || act_dim > 32) return 0;
  if (N < ORP_ROWS || N % ORP_ROWS != 0) return 0;
  return 1;
}

size_t osa_rollout_persistent_ws_doubles(int N, int obs_dim) {
  if (N < 1 || obs_dim < 1) return 0;
  const size_t nrb = (size_t)(N + ORP_ROWS - 1) / ORP_ROWS;
  return 2 * (nrb * obs_dim * 2 + nrb) + 1;  // two partial-sum buffers + {arrival counter, time-out flag}
}

int osa_rollout_persistent(const osa_rollout_desc* d, void* stream) {
  OSA_REQUIRE(d != nullptr);
  const int obs_dim = d->obs_dim, N = d->num_envs;
  if (!osa_rollout_persistent_supported(obs_dim, d->act_dim, d->hidden, N)) return OSA_EUNSUPPORTED;
  OSA_REQUIRE(d->params && d->obs && d->act && d->value_r && d->value_c && d->logp && d->reward && d->cost);
  OSA_REQUIRE(d->path_end && d->boot_r && d->boot_c && d->ep_done && d->ep_ret_out && d->ep_cost_out);
  OSA_REQUIRE(d->ep_len_out && d->ep_ret && d->ep_cost && d->ep_len && d->last_obs && d->final_norm && d->act_env);
  OSA_REQUIRE(d->old_min && d->old_max && d->vscratch && d->norm_mean && d->norm_sumsq && d->norm_var);
  OSA_REQUIRE(d->norm_std && d->norm_count && d->ws && d->env_steps && d->env_obs && d->env_final);
  OSA_REQUIRE(d->env_reward && d->env_cost && d->env_terminated && d->env_truncated);
  OSA_REQUIRE(d->steps > 0 && d->horizon > 0);
  OsaRollArgs a = {};
  a.nd = osa_make_net(obs_dim, d->act_dim, d->hidden);
  a.params = d->params; a.N = N; a.T = d->steps; a.D = obs_dim; a.A = d->act_dim;
  a.obs = d->obs; a.act = d->act; a.value_r = d->value_r; a.value_c = d->value_c; a.logp = d->logp;
  a.reward = d->reward; a.cost = d->cost; a.path_end = d->path_end; a.boot_r = d->boot_r; a.boot_c = d->boot_c;
  a.ep_done = d->ep_done; a.ep_ret_out = d->ep_ret_out; a.ep_cost_out = d->ep_cost_out; a.ep_len_out = d->ep_len_out;
  a.ep_ret = d->ep_ret; a.ep_cost = d->ep_cost; a.ep_len = d->ep_len; a.last_obs = d->last_obs;
  a.final_norm = d->final_norm; a.act_env = d->act_env; a.old_min = d->old_min; a.old_max = d->old_max;
  a.vscratch = d->vscratch; a.n_mean = d->norm_mean; a.n_sumsq = d->norm_sumsq; a.n_var = d->norm_var;
  a.n_std = d->norm_std; a.n_count = d->norm_count; a.clip = d->norm_clip;
  const int nrb = N / ORP_ROWS;
  const size_t nws = osa_rollout_persistent_ws_doubles(N, obs_dim);
  a.ws = d->ws;
  a.bar = reinterpret_cast<unsigned int*>(d->ws + (nws - 1));
  a.env_seed = d->env_seed; a.env_step0 = d->env_step; a.env_step_base = d->env_step_base; a.horizon = d->horizon;
  a.cost_p = d->cost_p; a.steps = d->env_steps; a.env_obs = d->env_obs; a.env_final = d->env_final;
  a.env_reward = d->env_reward; a.env_cost = d->env_cost; a.env_term = d->env_terminated;
  a.env_trunc = d->env_truncated;
  a.defer_critics = d->defer_critics;
  a.pi_seed = d->noise_seed; a.pi_off0 = d->noise_offset; a.pi_off_base = d->noise_offset_base;
  hipStream_t st = osa_stream(stream);
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return OSA_EHIP;
  if (nrb > cus) return OSA_EUNSUPPORTED;  // the workgroups meet at a grid barrier: one per compute unit at most
  // arrival counter back to 0 (the sticky flag in the upper half of the word pair stays)
  if (hipMemsetAsync(a.bar, 0, sizeof(unsigned int), st) != hipSuccess) return OSA_EHIP;
  size_t lds = (size_t)2 * ORP_ROWS * obs_dim * sizeof(float) + (size_t)ORP_STAGE_BLOCKS * obs_dim * 2 * sizeof(double);
  {  // the actor's padded parameter block, if the compute unit's 160 KB hold it next to the observation rows
    const size_t actor = ((size_t)a.nd.H * (a.nd.INP + 4) + (size_t)a.nd.H * (a.nd.H + 4) + (size_t)a.nd.OUTP * (a.nd.H + 4) +
                          2 * (size_t)a.nd.H + 2 * (size_t)a.nd.OUTP) * sizeof(float);
    const size_t budget = 160 * 1024 - 8 * 1024;  // (the kernel's static LDS: OrpShared, < 8 KB)
    a.actor_in_lds = (lds + actor <= budget && !osa_env_is_zero("OSA_ROLLOUT_ACTOR_LDS")) ? 1 : 0;
    if (a.actor_in_lds) lds += actor;
  }
  a.dbg = g_orp_clocks;
  const int OT = a.nd.OUTP / 16;
#define ORP_GO(O)                                                                                                   \
  do {                                                                                                             \
    static size_t attr_lds = 0;                                                                                    \
    if (lds > attr_lds) {                                                                                          \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&osa_rollout_persistent_kernel<4, O>),                 \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)                 \
        return OSA_EHIP;                                                                                           \
      attr_lds = lds;                                                                                              \
    }                                                                                                              \
    hipLaunchKernelGGL((osa_rollout_persistent_kernel<4, O>), dim3(nrb), dim3(512), lds, st, a);                   \
  } while (0)
  if (OT == 1) ORP_GO(1);
  else if (OT == 2) ORP_GO(2);
  else return OSA_EUNSUPPORTED;
#undef ORP_GO
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

// the sticky time-out flag of the grid barrier (a workgroup never arrived: results invalid)
int osa_rollout_persistent_timed_out(const double* ws, int N, int obs_dim, int* out) {
  OSA_REQUIRE(ws && out);
  const size_t nws = osa_rollout_persistent_ws_doubles(N, obs_dim);
  const unsigned int* bar = reinterpret_cast<const unsigned int*>(ws + (nws - 1));
  return hipMemcpy(out, bar + 1, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess ? OSA_OK : OSA_EHIP;
}

}  // extern "C"
