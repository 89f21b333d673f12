// ConstraintActorCritic.step (omnisafe/models/actor_critic/constraint_actor_critic.py:84-109) for the 16 rows of one
// wave and ONE network: forward pass (mlp_device.h), then -- actor -- sample a = mu + eps sigma (injected noise or
// Philox4x32-10 + Box-Muller, one counter block per (row, dimension)), log-probability, ActionScale (envs/wrapper.py:
// 510-514) and -- critics -- the value.  The body of osa_policy_step_kernel (mlp_kernels.hip: one launch per vector
// step).
#pragma once
#include "mlp_device.h"

template <int HT, int OT>
__device__ __forceinline__ void osa_policy_rows(
    const OsaNet& nd, const float* __restrict__ params, int net, const float* __restrict__ xrow, int ld, bool vec_ok,
    long row, bool valid, const float* __restrict__ eps, unsigned long long seed, unsigned long long offset,
    int deterministic, float* __restrict__ act, int ld_act, float* __restrict__ value_r,
    float* __restrict__ value_c, float* __restrict__ logp, float* __restrict__ mean_out, int ld_mean,
    float* __restrict__ act_env, int ld_env, const float* __restrict__ old_min, const float* __restrict__ old_max,
    float min_a, float max_a) {
  const int lane = threadIdx.x & 63, g = lane >> 4;
  const float* p = params + (long)net * nd.P;
  f32x4 h1[HT], h2[HT], out[OT];
  osa_mlp_forward<HT, OT>(nd, p, xrow, ld, vec_ok, h1, h2, out);
  if (net == 0) {
    float lp = 0.f;
#pragma unroll
    for (int o = 0; o < OT; ++o) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = 16 * o + 4 * g + r;
        if (d < nd.act_dim && valid) {
          const float mu = out[o][r];
          const float sd = expf(p[nd.oLS + d]);
          float a = mu;
          if (!deterministic) {
            float e;
            if (eps != nullptr) {
              e = eps[row * nd.act_dim + d];
            } else {
              uint32_t w[4];
              osa_philox(seed, offset, (unsigned long long)row * nd.act_dim + d, w);
              float e1;
              osa_box_muller(w[0], w[1], e, e1);
            }
            a = mu + e * sd;  // Normal.rsample: loc + eps * scale
          }
          if (act) act[row * ld_act + d] = a;
          // ActionScale.step of the wrapper chain in the same launch (osa_policy_step_scaled)
          if (act_env) act_env[row * ld_env + d] = osa_action_scale1(a, old_min[d], old_max[d], min_a, max_a);
          if (mean_out) mean_out[row * ld_mean + d] = mu;
          // Normal.log_prob: -((v - loc)^2) / (2 var) - log(scale) - log(sqrt(2 pi))
          const float z = a - mu;
          lp += -(z * z) / (2.f * (sd * sd)) - logf(sd) - 0.91893853320467274178f;
        }
      }
    }
    lp = osa_sum_over_groups(lp);
    if (g == 0 && valid && logp) logp[row] = lp;
  } else {
    float* __restrict__ dst = (net == 1) ? value_r : value_c;
    if (g == 0 && valid && dst) dst[row] = out[0][0];
  }
}
