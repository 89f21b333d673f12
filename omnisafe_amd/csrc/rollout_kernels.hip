// Rollout-side kernels: running observation normaliser (K2), action scaling (K3), per-step episode
// accounting + bootstrap selection (the per-env Python loop of OnPolicyAdapter.rollout), and the
// synthetic fixed-shape vector environment used by the throughput benchmark.
#include "mlp_device.h"

// ------------------------------------------------------------------------------------------------
// K2  Normalizer._push + normalize  (omnisafe/common/normalizer.py:88-139)
// State (device): mean[D], sumsq[D], var[D], std[D] float32; count int64[1].
// The batch moments are reduced in float64 as shifted sums S1 = sum(x - c), S2 = sum((x - c)^2) with
// c = current running mean, then merged with the reference's Chan update in float32.
// ------------------------------------------------------------------------------------------------
#define OSA_NORM_ROWS 128  // rows per workgroup in the partial reduction

// One launch: every workgroup reduces its 128 rows x 64 columns to float64 partial sums in `ws`, takes a
// ticket, and the LAST workgroup to arrive merges all partials into the running state (release fence ->
// agent-scope atomic -> acquire fence; the ticket is left at 0 for the next call).  Round 1 used two
// launches (partial, merge: 13 + 11.5 us per vector step for a 1 MB input).
// partial sums of the single-launch reduction: relaxed agent-scope 8-byte accesses (performed at / served by the
// device-coherent level, so that the last workgroup to arrive sees them whichever XCC wrote them)
__device__ __forceinline__ void osa_ws_put(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double osa_ws_get(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p),
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__global__ __launch_bounds__(256) void osa_norm_push_kernel(
    const float* __restrict__ x, int ld, int N, int D, const uint8_t* __restrict__ mask,
    float* __restrict__ mean, float* __restrict__ sumsq, float* __restrict__ var,
    float* __restrict__ std_, long* __restrict__ count, double* __restrict__ ws, int* __restrict__ ticket) {
#pragma clang fp contract(off)
  __shared__ double s1[4][64], s2[4][64];
  __shared__ int scnt[4];
  __shared__ int s_last;
  __shared__ long s_n;
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + cx;
  const int r0 = blockIdx.x * OSA_NORM_ROWS;
  const int r1 = min(N, r0 + OSA_NORM_ROWS);
  const int nrb = gridDim.x;
  {
    const double c = (col < D) ? (double)mean[col] : 0.0;
    double a1 = 0.0, a2 = 0.0;
    int cnt = 0;
    // (ALL of a thread's 32 rows requested before the first is consumed -- from clamped addresses, consumed in row
    // order: the row-by-row loop was a chain of 32 dependent memory round trips per thread, 25 us per call for 1 MB;
    // 8 rows per trip still were four round trips of a kernel that is nothing but latency)
    const int cl = (col < D) ? col : 0;
    float xv[OSA_NORM_ROWS / 4];
    uint8_t mv[OSA_NORM_ROWS / 4];
#pragma unroll
    for (int u = 0; u < OSA_NORM_ROWS / 4; ++u) {
      const int r = min(r0 + ry + 4 * u, r1 - 1);
      xv[u] = x[(long)r * ld + cl];
      mv[u] = mask ? mask[r] : (uint8_t)1;
    }
#pragma unroll
    for (int u = 0; u < OSA_NORM_ROWS / 4; ++u) {
      if (r0 + ry + 4 * u < r1 && mv[u] != 0) {
        ++cnt;
        if (col < D) {
          const double d = (double)xv[u] - c;
          a1 += d;
          a2 = __builtin_fma(d, d, a2);
        }
      }
    }
    s1[ry][cx] = a1;
    s2[ry][cx] = a2;
    if (cx == 0) scnt[ry] = cnt;
  }
  __syncthreads();
  if (ry == 0) {
    if (col < D) {
      double* o = ws + ((long)blockIdx.x * D + col) * 2;
      osa_ws_put(o, s1[0][cx] + s1[1][cx] + s1[2][cx] + s1[3][cx]);
      osa_ws_put(o + 1, s2[0][cx] + s2[1][cx] + s2[2][cx] + s2[3][cx]);
    }
    if (cx == 0 && blockIdx.y == 0)
      osa_ws_put(ws + (long)nrb * D * 2 + blockIdx.x, (double)(scnt[0] + scnt[1] + scnt[2] + scnt[3]));
  }
  // ---- last-arriver ticket.  The partials travel as agent-scope (write-through / cache-bypassing) 8-byte
  // accesses: "my stores are performed" is an s_waitcnt, and nobody needs an agent-scope RELEASE FENCE -- which
  // writes back every dirty line of the XCC's L2 and made this kernel cost 21 us even for 16 rows
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int total = gridDim.x * gridDim.y;
    const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == total - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  // ---- merge (Normalizer._push, normalizer.py:109-139): every other workgroup has finished reading `mean`
  if (threadIdx.x < 64) {  // the selected rows: a sum of small integers (exact in any order) -- one wave, all loads at once
    double n = 0.0;
    for (int b = threadIdx.x; b < nrb; b += 64) n += osa_ws_get(ws + (long)nrb * D * 2 + b);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) n += __shfl_xor(n, o);
    if (threadIdx.x == 0) {
      s_n = (long)n;
      __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next call
    }
  }
  __syncthreads();
  const long n_raw = s_n;
  if (n_raw == 0) return;
  const long cnt_old = *count;
  const long cnt_new = cnt_old + n_raw;
  for (int c0 = threadIdx.x; c0 < D; c0 += blockDim.x) {
    double S1 = 0.0, S2 = 0.0;
    for (int b0 = 0; b0 < nrb; b0 += 32) {  // 32 partials' loads in flight (4096 envs: all of them), summed in order
      double p1[32], p2[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const int b = min(b0 + u, nrb - 1);
        p1[u] = osa_ws_get(ws + ((long)b * D + c0) * 2);
        p2[u] = osa_ws_get(ws + ((long)b * D + c0) * 2 + 1);
      }
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        if (b0 + u < nrb) {
          S1 += p1[u];
          S2 += p2[u];
        }
      }
    }
    const double c = (double)mean[c0];
    const float mean_raw = (float)(c + S1 / (double)n_raw);
    // sum((x - mean_raw)^2) = S2 - S1^2/n   (exact identity; float64 keeps ~1e-12 relative)
    double q = S2 - S1 * S1 / (double)n_raw;
    if (q < 0.0) q = 0.0;
    const float sumq_raw = (float)q;
    float m_new, ss_new;
    if (cnt_old == 0) {  // normalizer.py:118-126 (first push)
      m_new = mean_raw;
      ss_new = sumq_raw;
    } else {  // normalizer.py:127-136
      const float delta = mean_raw - mean[c0];
      m_new = mean[c0] + delta * (float)n_raw / (float)cnt_new;
      ss_new = sumsq[c0] + (sumq_raw + delta * delta * (float)cnt_old * (float)n_raw / (float)cnt_new);
    }
    mean[c0] = m_new;
    sumsq[c0] = ss_new;
    const float v = ss_new / (float)(cnt_new - 1);  // count == 1 -> 0/0 = NaN, as the reference
    var[c0] = v;
    const float s = sqrtf(v);
    std_[c0] = fmaxf(s, 1e-2f);  // torch.max(std, 1e-2): NaN propagates like torch.max
    if (s != s) std_[c0] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) *count = cnt_new;
}

__global__ __launch_bounds__(256) void osa_normalize_kernel(
    const float* __restrict__ x, int ld_x, float* __restrict__ y, int ld_y, int N, int D,
    const uint8_t* __restrict__ mask, const float* __restrict__ mean, const float* __restrict__ std_,
    const long* __restrict__ count, float clip) {
#pragma clang fp contract(off)
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)N * D) return;
  const int r = (int)(gid / D), col = (int)(gid - (long)r * D);
  float v = x[(long)r * ld_x + col];
  const bool on = mask == nullptr || mask[r] != 0;
  if (on && *count > 1) {  // normalizer.py:104-107
    v = (v - mean[col]) / std_[col];
    v = fminf(fmaxf(v, -clip), clip);
  }
  y[(long)r * ld_y + col] = v;
}

// ------------------------------------------------------------------------------------------------
// K3  ActionScale.step (omnisafe/envs/wrapper.py:510-514)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void osa_action_scale_kernel(
    const float* __restrict__ act, int ld_a, float* __restrict__ out, int ld_o, int N, int D,
    const float* __restrict__ old_min, const float* __restrict__ old_max, float min_a, float max_a) {
#pragma clang fp contract(off)
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)N * D) return;
  const int r = (int)(gid / D), d = (int)(gid - (long)r * D);
  const float a = act[(long)r * ld_a + d];
  out[(long)r * ld_o + d] = osa_action_scale1(a, old_min[d], old_max[d], min_a, max_a);
}

// ------------------------------------------------------------------------------------------------
// a1  per-step episode accounting + bootstrap selection (onpolicy_adapter.py:86-136, :138-190)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void osa_rollout_post_step_kernel(
    int N, int epoch_end, const float* __restrict__ reward, const float* __restrict__ cost,
    const uint8_t* __restrict__ terminated, const uint8_t* __restrict__ truncated,
    const float* __restrict__ vnext_r, const float* __restrict__ vnext_c,
    const float* __restrict__ vfinal_r, const float* __restrict__ vfinal_c,
    float* __restrict__ ep_ret, float* __restrict__ ep_cost, float* __restrict__ ep_len,
    uint8_t* __restrict__ path_end, float* __restrict__ boot_r, float* __restrict__ boot_c,
    uint8_t* __restrict__ ep_done, float* __restrict__ ep_ret_out, float* __restrict__ ep_cost_out,
    float* __restrict__ ep_len_out, float* __restrict__ reward_row, float* __restrict__ cost_row) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  // buffer.store's reward / cost rows (:100-108) when the caller has no normaliser or adapter hook in between
  if (reward_row) reward_row[n] = reward[n];
  if (cost_row) cost_row[n] = cost[n];
  // _log_value (:155-157)
  float er = ep_ret[n] + reward[n];
  float ec = ep_cost[n] + cost[n];
  float el = ep_len[n] + 1.f;
  const bool done = terminated[n] != 0, time_out = truncated[n] != 0;
  uint8_t pe = 0, ed = 0;
  float lr = 0.f, lc = 0.f;
  if (epoch_end || done || time_out) {  // :115-136
    if (!done) {
      if (epoch_end && vnext_r) {
        lr = vnext_r[n];
        lc = vnext_c[n];
      }
      if (time_out && vfinal_r) {
        lr = vfinal_r[n];
        lc = vfinal_c[n];
      }
    }
    pe = 1;
    if (done || time_out) {
      ed = 1;
      ep_ret_out[n] = er;
      ep_cost_out[n] = ec;
      ep_len_out[n] = el;
      er = ec = el = 0.f;
    }
  }
  path_end[n] = pe;
  boot_r[n] = lr;
  boot_c[n] = lc;
  ep_done[n] = ed;
  ep_ret[n] = er;
  ep_cost[n] = ec;
  ep_len[n] = el;
}


// ------------------------------------------------------------------------------------------------
// Saute / Simmer state augmentation (omnisafe/adapter/saute_adapter.py:124-196): one thread per env.
//   _safety_step      z <- (z - cost / budget) / saute_gamma
//   _safety_reward    r <- r if z > 0 else unsafe_reward
//   episode end       z <- z * (1 - done) + done * z_reset
//   _augment_obs      column `col` of the next / final observation rows <- z   (final rows get the value
//                     AFTER the reset, as the reference augments them after updating _safety_obs)
//   _log_value / _log_metrics: ep_budget += z; a finished episode writes its sum to ep_budget_out
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void osa_saute_step_kernel(
    int N, const float* __restrict__ cost, const float* __restrict__ reward,
    const uint8_t* __restrict__ terminated, const uint8_t* __restrict__ truncated,
    float* __restrict__ safety_obs, const float* __restrict__ budget, float saute_gamma,
    float unsafe_reward, const float* __restrict__ reset_value, float* __restrict__ reward_out,
    float* __restrict__ next_rows, int ld_next, float* __restrict__ final_rows, int ld_final, int col,
    float* __restrict__ ep_budget, float* __restrict__ ep_budget_out) {
#pragma clang fp contract(off)
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float z = safety_obs[n];
  z = z - cost[n] / budget[n];
  z = z / saute_gamma;
  const float safe = z > 0.f ? 1.f : 0.f;
  reward_out[n] = safe * reward[n] + (1.f - safe) * unsafe_reward;
  const float done = (terminated[n] != 0 || truncated[n] != 0) ? 1.f : 0.f;
  z = z * (1.f - done) + done * reset_value[n];
  safety_obs[n] = z;
  next_rows[(long)n * ld_next + col] = z;
  if (final_rows) final_rows[(long)n * ld_final + col] = z;
  const float eb = ep_budget[n] + z;
  if (done != 0.f) {
    ep_budget_out[n] = eb;
    ep_budget[n] = 0.f;
  } else {
    ep_budget[n] = eb;
  }
}

// ------------------------------------------------------------------------------------------------
// Synthetic fixed-shape vector environment (benchmark stand-in for Safety-Gymnasium, whose MuJoCo
// physics is third-party CPU code outside the reference repository): obs ~ N(0,1)^D_o,
// reward ~ N(0,1), cost ~ Bernoulli(p), never terminates, truncates every `horizon` steps with the
// gymnasium vector auto-reset convention (returned obs is the post-reset obs, the pre-reset obs is
// handed back as final_observation).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void osa_synth_env_kernel(
    unsigned long long seed, unsigned long long step, const unsigned long long* __restrict__ step_base, int N,
    int D, int horizon, float cost_p, int* __restrict__ steps, float* __restrict__ obs, int ld,
    float* __restrict__ reward, float* __restrict__ cost, uint8_t* __restrict__ terminated,
    uint8_t* __restrict__ truncated, float* __restrict__ final_obs, int ld_f, int reset_only) {
  if (step_base) step += *step_base;  // device-resident part of the Philox stream position
  const int n = blockIdx.x;  // one workgroup per env row; threads over feature pairs
  uint8_t trunc = 0;
  if (!reset_only) trunc = (steps[n] + 1 >= horizon) ? 1 : 0;
  for (int pair = threadIdx.x; 2 * pair < D; pair += blockDim.x) {
    uint32_t w[4];
    osa_philox(seed, step, ((unsigned long long)n << 20) + pair, w);
    float a, b, c2, d2;
    osa_box_muller(w[0], w[1], a, b);
    osa_box_muller(w[2], w[3], c2, d2);  // second pair: the post-reset observation
    const int i0 = 2 * pair, i1 = 2 * pair + 1;
    if (trunc) {
      if (final_obs) {
        final_obs[(long)n * ld_f + i0] = a;
        if (i1 < D) final_obs[(long)n * ld_f + i1] = b;
      }
      obs[(long)n * ld + i0] = c2;
      if (i1 < D) obs[(long)n * ld + i1] = d2;
    } else {
      obs[(long)n * ld + i0] = a;
      if (i1 < D) obs[(long)n * ld + i1] = b;
    }
  }
  if (threadIdx.x == 0 && !reset_only) {
    uint32_t w[4];
    osa_philox(seed ^ 0x9E3779B97F4A7C15ull, step, (unsigned long long)n, w);
    float a, b;
    osa_box_muller(w[0], w[1], a, b);
    reward[n] = a;
    cost[n] = (osa_u01(w[2]) <= cost_p) ? 1.f : 0.f;
    terminated[n] = 0;
    truncated[n] = trunc;
    steps[n] = trunc ? 0 : steps[n] + 1;
  }
  if (threadIdx.x == 0 && reset_only) steps[n] = 0;
}

// ------------------------------------------------------------------------------------------------
// Learnable synthetic CMDP "SynthReach" (point reaches resampled goals, one static hazard disc):
// state[n][8] = p(2) g(2) h(2) pad(2).  Dynamics stated in oracle/np_oracle.py:reach_env_step and
// compared with it by tests/test_learning_gpu.py; float32, no fused multiply-adds so that the CPU twin
// the reference trains on sees the same arithmetic.  One wave per env: every lane computes the (tiny)
// transition, lane k writes column k of the observation row (coalesced 240-byte row stores).
// ------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
__device__ __forceinline__ float osa_reach_dist(float ux, float uy, float vx, float vy) {
  const float dx = ux - vx, dy = uy - vy;
  const float xx = dx * dx, yy = dy * dy;
  return sqrtf(xx + yy);
}

__device__ __forceinline__ float osa_reach_obs_col(const float (&s)[6], int k) {
  float v = 0.f;
  if (k == 0) v = s[0];
  if (k == 1) v = s[1];
  if (k == 2) v = s[2] - s[0];
  if (k == 3) v = s[3] - s[1];
  if (k == 4) v = s[4] - s[0];
  if (k == 5) v = s[5] - s[1];
  return v;
}

__device__ __forceinline__ float osa_reach_uniform(uint32_t w) {  // [-1, 1]
  return 2.f * osa_u01(w) - 1.f;
}

__global__ __launch_bounds__(64) void osa_reach_env_kernel(
    unsigned long long seed, unsigned long long step, const unsigned long long* __restrict__ step_base, int N,
    int D, int horizon, float* __restrict__ state, int* __restrict__ steps,
    const float* __restrict__ action, int ld_a,
    float* __restrict__ obs, int ld, float* __restrict__ reward, float* __restrict__ cost,
    uint8_t* __restrict__ terminated, uint8_t* __restrict__ truncated, float* __restrict__ final_obs,
    int ld_f, int reset_only) {
  if (step_base) step += *step_base;
  const int n = blockIdx.x, lane = threadIdx.x;
  float s[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) s[k] = state[(long)n * 8 + k];
  uint32_t w0[4], w1[4];  // fresh positions for a reset: 6 uniforms
  osa_philox(seed ^ 0xD1B54A32D192ED03ull, step, ((unsigned long long)n << 20) + 1, w0);
  osa_philox(seed ^ 0xD1B54A32D192ED03ull, step, ((unsigned long long)n << 20) + 2, w1);
  const float fresh[6] = {osa_reach_uniform(w0[0]), osa_reach_uniform(w0[1]), osa_reach_uniform(w0[2]),
                          osa_reach_uniform(w0[3]), osa_reach_uniform(w1[0]), osa_reach_uniform(w1[1])};
  uint8_t trunc = 0;
  float r = 0.f, c = 0.f;
  if (reset_only) {
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] = fresh[k];
  } else {
    const float a0 = fminf(fmaxf(action[(long)n * ld_a + 0], -1.f), 1.f);
    const float a1 = fminf(fmaxf(action[(long)n * ld_a + 1], -1.f), 1.f);
    const float m0 = 0.1f * a0, m1 = 0.1f * a1;
    const float qx = fminf(fmaxf(s[0] + m0, -1.5f), 1.5f);
    const float qy = fminf(fmaxf(s[1] + m1, -1.5f), 1.5f);
    const float d0 = osa_reach_dist(s[0], s[1], s[2], s[3]);
    const float d1 = osa_reach_dist(qx, qy, s[2], s[3]);
    const bool reached = d1 < 0.15f;
    r = (d0 - d1) + (reached ? 1.f : 0.f);
    c = (osa_reach_dist(qx, qy, s[4], s[5]) < 0.3f) ? 1.f : 0.f;
    s[0] = qx;
    s[1] = qy;
    if (reached) {
      s[2] = osa_reach_uniform(w1[2]);
      s[3] = osa_reach_uniform(w1[3]);
    }
    trunc = (steps[n] + 1 >= horizon) ? 1 : 0;
    if (trunc) {
      if (final_obs)
        for (int k = lane; k < D; k += 64) final_obs[(long)n * ld_f + k] = osa_reach_obs_col(s, k);
#pragma unroll
      for (int k = 0; k < 6; ++k) s[k] = fresh[k];
    }
  }
  for (int k = lane; k < D; k += 64) obs[(long)n * ld + k] = osa_reach_obs_col(s, k);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) state[(long)n * 8 + k] = s[k];
    if (reset_only) {
      steps[n] = 0;
    } else {
      reward[n] = r;
      cost[n] = c;
      terminated[n] = 0;
      truncated[n] = trunc;
      steps[n] = trunc ? 0 : steps[n] + 1;
    }
  }
}
#pragma clang fp contract(fast)

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// End-of-epoch log flush (onpolicy_adapter.py:159-174 _log_metrics, :88-92 logger.store Value/*): the finished
// episodes of the epoch compacted in (step, env) order -- flat index, return, cost, length and an optional extra
// column -- and the means of the two value columns.  Rounds 1-3 used torch for this (nonzero = rocPRIM partition +
// scan, three gathers, two means, a stack: a dozen launches); here two: (1) every workgroup counts the finished
// episodes of its contiguous range and sums its share of the value columns in float64, the LAST to arrive scans the
// counts into offsets and finishes the means; (2) every workgroup compacts its range behind its offset.
// ws (layout independent of the launch's grid, so that one zero-initialised workspace serves any M): ints [0] ticket
// (left at 0 by every call), [1 .. 256] counts, [257 .. 512] offsets; doubles from OSA_FLUSH_DOFF: [2 G] value sums.
// ------------------------------------------------------------------------------------------------
#define OSA_FLUSH_MAXG 256
#define OSA_FLUSH_DOFF 260  // doubles
__global__ __launch_bounds__(256) void osa_flush_count_kernel(const uint8_t* __restrict__ done, long M,
                                                              const float* __restrict__ value_r,
                                                              const float* __restrict__ value_c, double* __restrict__ ws,
                                                              int* __restrict__ out_count, float* __restrict__ out_means) {
  __shared__ double red[17];
  __shared__ int sc[4];
  __shared__ int s_last;
  const int G = gridDim.x, b = blockIdx.x;
  int* iw = reinterpret_cast<int*>(ws);
  double* dw = ws + OSA_FLUSH_DOFF;
  const long per = (M + G - 1) / G, lo = (long)b * per, hi = min(M, lo + per);
  int cnt = 0;
  double sr = 0.0, sv = 0.0;
  for (long i0 = lo + threadIdx.x; i0 < hi; i0 += 256 * 8) {  // 8 x 3 loads in flight, consumed in order
    uint8_t f[8];
    float a[8], c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long i = min(i0 + 256L * u, hi - 1);
      f[u] = done[i];
      a[u] = value_r[i];
      c[u] = value_c[i];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (i0 + 256L * u < hi) {
        cnt += f[u] != 0;
        sr += (double)a[u];
        sv += (double)c[u];
      }
    }
  }
  sr = osa_block_sum<256>(sr, red);
  sv = osa_block_sum<256>(sv, red);
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
  if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    osa_ws_put(dw + 2 * b, sr);
    osa_ws_put(dw + 2 * b + 1, sv);
    __hip_atomic_store(iw + 1 + b, sc[0] + sc[1] + sc[2] + sc[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int t = __hip_atomic_fetch_add(iw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == G - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  // ---- the last workgroup to arrive: counts -> offsets (block order: nothing depends on the arrival order), means.
  // Every thread fetches one workgroup's partials (one round trip for all), thread 0 scans them in LDS.
  __shared__ int l_cnt[OSA_FLUSH_MAXG];
  __shared__ double l_sum[2][OSA_FLUSH_MAXG];
  for (int k = threadIdx.x; k < G; k += 256) {
    l_cnt[k] = __hip_atomic_load(iw + 1 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    l_sum[0][k] = osa_ws_get(dw + 2 * k);
    l_sum[1][k] = osa_ws_get(dw + 2 * k + 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    double tr = 0.0, tv = 0.0;
    for (int k = 0; k < G; ++k) {
      const int c = l_cnt[k];
      l_cnt[k] = run;
      run += c;
      tr += l_sum[0][k];
      tv += l_sum[1][k];
    }
    *out_count = run;
    out_means[0] = (float)(tr / (double)M);
    out_means[1] = (float)(tv / (double)M);
    __hip_atomic_store(iw, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next call
  }
  __syncthreads();
  for (int k = threadIdx.x; k < G; k += 256) iw[1 + OSA_FLUSH_MAXG + k] = l_cnt[k];
}

__global__ __launch_bounds__(256) void osa_flush_compact_kernel(const uint8_t* __restrict__ done,
                                                                const float* __restrict__ ep_ret,
                                                                const float* __restrict__ ep_cost,
                                                                const float* __restrict__ ep_len,
                                                                const float* __restrict__ extra, long M,
                                                                const double* __restrict__ ws, int* __restrict__ out_idx,
                                                                float* __restrict__ out_vals) {
  __shared__ int s_cnt[4];
  const int G = gridDim.x, b = blockIdx.x;
  const int* iw = reinterpret_cast<const int*>(ws);
  const long per = (M + G - 1) / G, lo = (long)b * per, hi = min(M, lo + per);
  // thread t: the contiguous sub-range [lo + t q, lo + (t + 1) q) of the workgroup's range
  const long q = (per + 255) / 256, tlo = min(hi, lo + (long)threadIdx.x * q), thi = min(hi, tlo + q);
  int cnt = 0;
  for (long i0 = tlo; i0 < thi; i0 += 16) {
    uint8_t f[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) f[u] = done[min(i0 + u, thi - 1)];
#pragma unroll
    for (int u = 0; u < 16; ++u) cnt += (i0 + u < thi) && f[u] != 0;
  }
  // exclusive scan of the 256 counts: inclusive scan inside each wave, then the waves' totals
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(inc, off, 64);
    if (lane >= off) inc += v;
  }
  if (lane == 63) s_cnt[wave] = inc;
  __syncthreads();
  int pos = iw[1 + OSA_FLUSH_MAXG + b] + inc - cnt;
  for (int w = 0; w < wave; ++w) pos += s_cnt[w];
  if (cnt == 0) return;
  for (long i0 = tlo; i0 < thi; i0 += 16) {
    uint8_t f[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) f[u] = done[min(i0 + u, thi - 1)];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const long i = i0 + u;
      if (i < thi && f[u] != 0) {
        out_idx[pos] = (int)i;
        out_vals[pos] = ep_ret[i];
        out_vals[M + pos] = ep_cost[i];
        out_vals[2 * M + pos] = ep_len[i];
        if (extra) out_vals[3 * M + pos] = extra[i];
        ++pos;
      }
    }
  }
}

// mean of x[idx[0 .. n)] (Value/Adv of the last minibatch, policy_gradient.py:369-377, 402): one workgroup, float64
__global__ __launch_bounds__(1024) void osa_gather_mean_kernel(const float* __restrict__ x, const long* __restrict__ idx,
                                                               long n, float* __restrict__ out) {
  __shared__ double red[17];
  double s = 0.0;
  for (long i0 = threadIdx.x; i0 < n; i0 += 1024 * 8) {  // indices, then values: two round trips per 8 elements
    long j[8];
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long i = min(i0 + 1024L * u, n - 1);
      j[u] = idx ? idx[i] : i;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x[j[u]];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + 1024L * u < n) s += (double)v[u];
  }
  s = osa_block_sum<1024>(s, red);
  if (threadIdx.x == 0) *out = (float)(s / (double)n);
}

extern "C" {

size_t osa_normalizer_ws_doubles(int N, int D) {
  if (N < 1 || D < 1) return 0;
  const size_t nrb = (size_t)(N + OSA_NORM_ROWS - 1) / OSA_NORM_ROWS;
  return 1 + nrb * D * 2 + nrb;  // ticket word (first, zero-initialised by the caller), partial sums, row counts
}

int osa_normalizer_push(const float* x, int ld, int N, int D, const uint8_t* mask, float* mean,
                        float* sumsq, float* var, float* std_, long* count, double* ws,
                        void* stream) {
  OSA_REQUIRE(x && mean && sumsq && var && std_ && count && ws && N > 0 && D > 0 && ld >= D);
  const int nrb = (N + OSA_NORM_ROWS - 1) / OSA_NORM_ROWS;
  // the ticket word is ws[0], the partials follow: its position must NOT depend on N -- a workspace that served
  // a large batch is reused for smaller ones (Normalizer._workspace), and a ticket behind the partials would
  // then land on a leftover partial sum
  int* ticket = reinterpret_cast<int*>(ws);
  hipLaunchKernelGGL(osa_norm_push_kernel, dim3(nrb, (D + 63) / 64), dim3(256), 0, osa_stream(stream), x, ld,
                     N, D, mask, mean, sumsq, var, std_, count, ws + 1, ticket);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_normalizer_apply(const float* x, int ld_x, float* y, int ld_y, int N, int D,
                         const uint8_t* mask, const float* mean, const float* std_, const long* count,
                         float clip, void* stream) {
  OSA_REQUIRE(x && y && mean && std_ && count && N > 0 && D > 0 && ld_x >= D && ld_y >= D);
  const long total = (long)N * D;
  hipLaunchKernelGGL(osa_normalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     osa_stream(stream), x, ld_x, y, ld_y, N, D, mask, mean, std_, count, clip);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_action_scale(const float* act, int ld_act, float* out, int ld_out, int N, int act_dim,
                     const float* old_min, const float* old_max, float min_action, float max_action,
                     void* stream) {
  OSA_REQUIRE(act && out && old_min && old_max && N > 0 && act_dim > 0);
  const long total = (long)N * act_dim;
  hipLaunchKernelGGL(osa_action_scale_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     osa_stream(stream), act, ld_act, out, ld_out, N, act_dim, old_min, old_max,
                     min_action, max_action);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_rollout_post_step(int N, int epoch_end, const float* reward, const float* cost,
                          const uint8_t* terminated, const uint8_t* truncated, const float* vnext_r,
                          const float* vnext_c, const float* vfinal_r, const float* vfinal_c,
                          float* ep_ret, float* ep_cost, float* ep_len, uint8_t* path_end,
                          float* boot_r, float* boot_c, uint8_t* ep_done, float* ep_ret_out,
                          float* ep_cost_out, float* ep_len_out, float* reward_row, float* cost_row,
                          void* stream) {
  OSA_REQUIRE(N > 0 && reward && cost && terminated && truncated && ep_ret && ep_cost && ep_len);
  OSA_REQUIRE(path_end && boot_r && boot_c && ep_done && ep_ret_out && ep_cost_out && ep_len_out);
  OSA_REQUIRE((vnext_r == nullptr) == (vnext_c == nullptr));
  OSA_REQUIRE((vfinal_r == nullptr) == (vfinal_c == nullptr));
  hipLaunchKernelGGL(osa_rollout_post_step_kernel, dim3((N + 255) / 256), dim3(256), 0,
                     osa_stream(stream), N, epoch_end, reward, cost, terminated, truncated, vnext_r,
                     vnext_c, vfinal_r, vfinal_c, ep_ret, ep_cost, ep_len, path_end, boot_r, boot_c,
                     ep_done, ep_ret_out, ep_cost_out, ep_len_out, reward_row, cost_row);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_saute_step(int N, const float* cost, const float* reward, const uint8_t* terminated,
                   const uint8_t* truncated, float* safety_obs, const float* budget, float saute_gamma,
                   float unsafe_reward, const float* reset_value, float* reward_out, float* next_rows,
                   int ld_next, float* final_rows, int ld_final, int col, float* ep_budget,
                   float* ep_budget_out, void* stream) {
  OSA_REQUIRE(N > 0 && cost && reward && terminated && truncated && safety_obs && budget && reset_value);
  OSA_REQUIRE(reward_out && next_rows && ld_next > col && col >= 0 && ep_budget && ep_budget_out);
  OSA_REQUIRE(!final_rows || ld_final > col);
  hipLaunchKernelGGL(osa_saute_step_kernel, dim3((N + 255) / 256), dim3(256), 0, osa_stream(stream), N,
                     cost, reward, terminated, truncated, safety_obs, budget, saute_gamma, unsafe_reward,
                     reset_value, reward_out, next_rows, ld_next, final_rows, ld_final, col, ep_budget,
                     ep_budget_out);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_synth_env_step(unsigned long long seed, unsigned long long step,
                       const unsigned long long* step_base, int N, int obs_dim,
                       int horizon, float cost_p, int* steps, float* obs, int ld_obs, float* reward,
                       float* cost, uint8_t* terminated, uint8_t* truncated, float* final_obs,
                       int ld_final, int reset_only, void* stream) {
  OSA_REQUIRE(N > 0 && obs_dim > 0 && steps && obs && ld_obs >= obs_dim);
  if (!reset_only) OSA_REQUIRE(reward && cost && terminated && truncated && horizon > 0);
  hipLaunchKernelGGL(osa_synth_env_kernel, dim3(N), dim3(64), 0, osa_stream(stream), seed, step, step_base, N,
                     obs_dim, horizon, cost_p, steps, obs, ld_obs, reward, cost, terminated,
                     truncated, final_obs, ld_final, reset_only);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_reach_env_step(unsigned long long seed, unsigned long long step,
                       const unsigned long long* step_base, int N, int obs_dim,
                       int horizon, float* state, int* steps, const float* action, int ld_action,
                       float* obs, int ld_obs, float* reward, float* cost, uint8_t* terminated,
                       uint8_t* truncated, float* final_obs, int ld_final, int reset_only,
                       void* stream) {
  OSA_REQUIRE(N > 0 && obs_dim >= 6 && state && steps && obs && ld_obs >= obs_dim);
  if (!reset_only)
    OSA_REQUIRE(action && ld_action >= 2 && reward && cost && terminated && truncated && horizon > 0);
  hipLaunchKernelGGL(osa_reach_env_kernel, dim3(N), dim3(64), 0, osa_stream(stream), seed, step, step_base, N,
                     obs_dim, horizon, state, steps, action, ld_action, obs, ld_obs, reward, cost,
                     terminated, truncated, final_obs, ld_final, reset_only);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}


size_t osa_episode_flush_ws_doubles(long M) {
  (void)M;
  return (size_t)OSA_FLUSH_DOFF + 2 * OSA_FLUSH_MAXG;
}

int osa_episode_flush(const uint8_t* done, const float* ep_ret, const float* ep_cost, const float* ep_len,
                      const float* extra, long M, const float* value_r, const float* value_c, int* out_count,
                      int* out_idx, float* out_vals, float* out_means, double* ws, void* stream) {
  OSA_REQUIRE(done && ep_ret && ep_cost && ep_len && value_r && value_c && M > 0);
  OSA_REQUIRE(out_count && out_idx && out_vals && out_means && ws);
  if (M >= 2147483647L) return OSA_EUNSUPPORTED;
  int G = (int)((M + 1023) / 1024);  // (1024 slots per workgroup: four per thread at the headline's 65 536)
  if (G > OSA_FLUSH_MAXG) G = OSA_FLUSH_MAXG;
  hipLaunchKernelGGL(osa_flush_count_kernel, dim3(G), dim3(256), 0, osa_stream(stream), done, M, value_r, value_c, ws,
                     out_count, out_means);
  hipLaunchKernelGGL(osa_flush_compact_kernel, dim3(G), dim3(256), 0, osa_stream(stream), done, ep_ret, ep_cost,
                     ep_len, extra, M, ws, out_idx, out_vals);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_gather_mean(const float* x, const long* idx, long n, float* out, void* stream) {
  OSA_REQUIRE(x && out && n > 0);
  hipLaunchKernelGGL(osa_gather_mean_kernel, dim3(1), dim3(1024), 0, osa_stream(stream), x, idx, n, out);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

}  // extern "C"
