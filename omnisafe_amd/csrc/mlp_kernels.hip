// Actor-critic kernels on gfx950 matrix cores: rollout policy step (K1), PPO-Lag minibatch
// forward/backward + gradient-norm clip + Adam (K9/K10), full-batch KL (K11).
// See mlp_device.h for the fragment algebra.  One workgroup = 256 threads = 4 waves = 64 samples;
// blockIdx.y selects the network (0 actor, 1 reward critic, 2 cost critic) so the three
// independent networks of ConstraintActorCritic run concurrently in one launch.
#include "mlp_device.h"
#include "policy_rows.h"

#define OSA_NSTAT 16

struct OsaHp {  // mirrors osa_ppo_hparams in include/omnisafe_amd.h
  float clip, entropy_coef, critic_norm_coef, max_grad_norm;
  float lr_actor, lr_critic, beta1, beta2, adam_eps;
  int use_critic_norm, use_max_grad_norm, use_cost;
  const float* lr_dev;  // optional {lr_actor, lr_critic} in device memory (osa_ppo_hparams.lr_device)
};

struct OsaMbArgs {
  OsaNet nd;
  float* params;   // [3][P]
  float* adam_m;   // [3][P]
  float* adam_v;   // [3][P]
  int* adam_step;  // [3]
  float* grads;    // [3][P]
  const float* obs;
  int ld_obs;
  const float* act;
  int ld_act;
  const float* logp;
  const float* tgt_r;
  const float* tgt_c;
  const float* adv_r;
  const float* adv_c;
  const long* idx;  // [B] sample rows, nullptr = identity
  int B;
  const float* lagrange;  // device scalar
  OsaHp hp;
  int mode;      // 0: grad + clip + Adam; 1: grad + clip only (all-reduce follows); 2: grad only
  int nblk;      // row blocks per network
  float* slabs;  // [3][nblk][P + OSA_NSTAT] partial gradients when nblk > 1
  float* stats;  // [OSA_NSTAT] statistics of this optimiser step
  int loss_kind; // 0 PPO clipped surrogate (base/ppo.py:66-78), 1 plain ratio*adv (policy_gradient.py:574)
  int nets_mask; // bit0 actor, bit1 reward critic, bit2 cost critic
  long long* dbg;  // optional [3][16] phase timestamps (s_memtime) of the last launch, or nullptr
  const float* vec;  // loss_kind 2 (Fisher-vector product): tangent vector, padded actor layout [P]
  float fvp_scale;   // 1 / (M * act_dim): natural_pg.py:95 takes .mean() over all M x D_a elements
  // extended surrogates (osa_ppo_minibatch_ext): per-sample KL(pi_theta || pi_old) term with optional
  // trust mask (FOCOPS focops.py:83-92, CUP cup.py:96-103) and P3O's exact-penalty term (p3o.py:62-68)
  const float* old_mean;     // [rows][ld_old_mean] or nullptr
  int ld_old_mean;
  const float* old_log_std;  // [act_dim]
  float ext_kl_coef;         // weight of the KL term (0 = off)
  float ext_mask_eta;        // >= 0: per-sample loss * 1[KL <= eta]; < 0: off
  float ext_ratio_scale;     // multiplies the ratio * adv surrogate term
  float ext_cost_kappa;      // > 0: + kappa * relu(mean(ratio * adv_c) + ext_cost_excess); single chunk only
  float ext_cost_excess;
  // slabs actually written per network when they differ from nblk (the balanced partial-gradient launch of the
  // large-batch step: nblk is then the slab STRIDE); nslab[0] < 0: every network has nblk slabs
  int nslab[3];
};

#define OSA_TICK(k)                                                                  \
  do {                                                                               \
    if (a.dbg && threadIdx.x == 0 && blockIdx.x == 0) a.dbg[blockIdx.y * 16 + (k)] = clock64(); \
  } while (0)

// ------------------------------------------------------------------------------------------------
// K1  rollout policy step
// ------------------------------------------------------------------------------------------------
template <int HT, int OT>
__global__ __launch_bounds__(256) void osa_policy_step_kernel(
    OsaNet nd, const float* __restrict__ params, const float* __restrict__ obs, int ld, int N,
    const float* __restrict__ eps, unsigned long long seed, unsigned long long offset,
    const unsigned long long* __restrict__ offset_base, int deterministic, int nets_mask,
    float* __restrict__ act, int ld_act, float* __restrict__ value_r, float* __restrict__ value_c,
    float* __restrict__ logp, float* __restrict__ mean_out, int ld_mean, float* __restrict__ act_env, int ld_env,
    const float* __restrict__ old_min, const float* __restrict__ old_max, float min_a, float max_a) {
  const int net = blockIdx.y;
  if (!((nets_mask >> net) & 1)) return;
  if (offset_base) offset += *offset_base;  // device-resident part of the Philox stream position
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15;
  const long row = (long)blockIdx.x * 64 + 16 * wave + j;
  const bool valid = row < N;
  const bool vec_ok = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(obs) & 15) == 0);
  // (the rows' arithmetic lives in policy_rows.h: the persistent rollout kernel runs the same function)
  osa_policy_rows<HT, OT>(nd, params, net, valid ? obs + row * ld : nullptr, ld, vec_ok, row, valid, eps, seed, offset,
                          deterministic, act, ld_act, value_r, value_c, logp, mean_out, ld_mean, act_env, ld_env,
                          old_min, old_max, min_a, max_a);
}

// ------------------------------------------------------------------------------------------------
// K9/K10  minibatch forward + backward (+ fused clip / Adam when one row block covers the batch)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float osa_block_sum_f(float v, float* red) {
  // deterministic block-wide float sum (any multiple-of-64 block size up to 1024); result to all
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  v = osa_wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += red[w];
    red[16] = s;
  }
  __syncthreads();
  return red[16];
}

// Gradient finalisation for one network, executed by one whole workgroup:
//   critic: g += 2*coef*p (the `param.pow(2).sum() * coef` term, policy_gradient.py:431-433)
//   total_norm = ||g||_2; g *= min(1, max_norm / (total_norm + 1e-6))   (clip_grad_norm_, :437-441)
//   Adam (torch.optim.Adam single-tensor step, bias-corrected), actor_critic.py:91-113
// Reads/writes grads[net] in place.  mode 0: clip + Adam, 1: clip only, 2: nothing (raw grads),
// 3: Adam only (gradients already clipped and averaged across ranks).
__device__ void osa_finalize_net(const OsaMbArgs& a, int net, float* red) {
  const OsaNet& nd = a.nd;
  const int P = nd.P;
  float* __restrict__ p = a.params + (long)net * P;
  float* __restrict__ gbuf = a.grads + (long)net * P;
  const bool critic = net != 0;
  const int mode = a.mode;
  float gclip = 1.f;
  if (mode != 3) {
    const bool l2 = critic && a.hp.use_critic_norm;
    const float c2 = 2.f * a.hp.critic_norm_coef;
    float gsq = 0.f, psq = 0.f;
    for (int e0 = 0; e0 < P; e0 += 8 * blockDim.x) {
      float g8[8], p8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int e = e0 + k * blockDim.x + threadIdx.x;
        if (e < P) {
          g8[k] = gbuf[e];
          p8[k] = p[e];
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int e = e0 + k * blockDim.x + threadIdx.x;
        if (e < P) {
          float gv = g8[k];
          const float pv = p8[k];
          if (critic && e >= nd.oLS) {  // critics have no log_std; keep the slot inert
            gv = 0.f;
          } else {
            if (l2) gv += c2 * pv;
            if (critic) psq += pv * pv;
          }
          gsq += gv * gv;
          gbuf[e] = gv;
        }
      }
    }
    OSA_TICK(9);
    gsq = osa_block_sum_f(gsq, red);
    psq = osa_block_sum_f(psq, red);
    const float total_norm = sqrtf(gsq);
    if (threadIdx.x == 0) {
      a.stats[7 + net] = total_norm;
      if (critic) a.stats[4 + net] = psq;  // [5] reward critic, [6] cost critic
    }
    if (mode == 2) return;
    if (a.hp.use_max_grad_norm) {
      float coef = a.hp.max_grad_norm / (total_norm + 1e-6f);
      coef = coef > 1.f ? 1.f : coef;
      if (mode == 1) {  // the clipped gradient itself is the result (all-reduce follows)
        for (int e = threadIdx.x; e < P; e += blockDim.x) gbuf[e] *= coef;
      } else {
        gclip = coef;   // mode 0: folded into the Adam pass below (one pass over global memory less)
      }
    }
    if (mode == 1) return;
    __syncthreads();
  }
  OSA_TICK(10);
  // ---- Adam
  const int step = a.adam_step[net] + 1;
  const double b1 = a.hp.beta1, b2 = a.hp.beta2;
  const double bc1 = 1.0 - pow(b1, (double)step);
  const double bc2 = 1.0 - pow(b2, (double)step);
  const float lr = a.hp.lr_dev ? a.hp.lr_dev[critic ? 1 : 0] : (critic ? a.hp.lr_critic : a.hp.lr_actor);
  const float step_size = (float)((double)lr / bc1);
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  OSA_TICK(11);
  const float beta1 = a.hp.beta1, beta2 = a.hp.beta2, eps = a.hp.adam_eps;
  float* __restrict__ m = a.adam_m + (long)net * P;
  float* __restrict__ v = a.adam_v + (long)net * P;
  for (int e0 = 0; e0 < P; e0 += 8 * blockDim.x) {  // batches of 8 independent loads per thread
    float gv8[8], mv8[8], vv8[8], pv8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int e = e0 + k * blockDim.x + threadIdx.x;
      if (e < P) {
        gv8[k] = gbuf[e];
        mv8[k] = m[e];
        vv8[k] = v[e];
        pv8[k] = p[e];
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int e = e0 + k * blockDim.x + threadIdx.x;
      if (e < P) {
        p[e] = osa_adam_update(gv8[k] * gclip, mv8[k], vv8[k], pv8[k], beta1, beta2, step_size, inv_bc2_sqrt, eps);
        m[e] = mv8[k];
        v[e] = vv8[k];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) a.adam_step[net] = step;
}

// HT = hidden width / 16 (2, 4, 8, 16: hidden_sizes [32, 32] ... [256, 256]); NSB = 16-sample blocks per chunk: 4 (64
// samples, one per lane of the four waves' S layout) while the four [H][SPC + 4] tiles fit the LDS, 2 at width 256
// (32 samples: waves 2, 3 idle through forward / backward, all four waves share the weight-gradient tiles).
template <int HT, int OT, int NSB>
__global__ __launch_bounds__(256) void osa_mb_grad_kernel(OsaMbArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int SPC = 16 * NSB;   // samples per chunk
  constexpr int OSA_SLD = SPC + 4;  // LDS leading dimension (floats) of the [feature][sample] transposed tiles
  const OsaNet& nd = a.nd;
  const int net = blockIdx.y;
  if (!((a.nets_mask >> net) & 1)) return;
  const int H = nd.H, INP = nd.INP, OUTP = nd.OUTP, P = nd.P;
  float* sH1 = reinterpret_cast<float*>(smem_raw);    // [H][SLD]
  float* sH2 = sH1 + H * OSA_SLD;                     // [H][SLD]
  float* sZ1 = sH2 + H * OSA_SLD;                     // [H][SLD]  dL/d(pre-activation 1)
  float* sZ2 = sZ1 + H * OSA_SLD;                     // [H][SLD]
  float* sDO = sZ2 + H * OSA_SLD;                     // [OUTP][SLD]  dL/d(output)
  float* sDL = sDO + OUTP * OSA_SLD;                  // [OUTP][SLD]  per-sample dL/d(log_std)
  float* red = sDL + OUTP * OSA_SLD;                  // [32]
  long* sIdx = reinterpret_cast<long*>(red + 32);     // [SPC] sample row or -1

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int i = j;  // the same lane bits index weight rows (A operand) and samples (B operand)
  const float* __restrict__ p = a.params + (long)net * P;
  const bool vec_ok = (a.ld_obs % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.obs) & 15) == 0);
  const float invB = 1.f / (float)a.B;
  float* __restrict__ gout =
      (a.nblk > 1) ? a.slabs + ((long)net * a.nblk + blockIdx.x) * (P + OSA_NSTAT) : a.grads + (long)net * P;
  float lam = 0.f;
  if (net == 0 && a.lagrange) lam = *a.lagrange;

  OSA_TICK(0);
  float loss_acc = 0.f, ratio_acc = 0.f;  // per-lane partial sums over this block's chunks
  const int nchunk = (a.B + SPC - 1) / SPC;
  const bool active = wave < NSB;  // this wave holds 16 samples of the chunk (S layout)
  bool first = true;
  for (int chunk = blockIdx.x; chunk < nchunk; chunk += a.nblk, first = false) {
    const int pos = chunk * SPC + 16 * wave + j;
    const bool valid = active && pos < a.B;
    const long row = valid ? (a.idx ? a.idx[pos] : (long)pos) : -1;
    __syncthreads();  // previous chunk's LDS fully consumed
    if (g == 0 && active) sIdx[16 * wave + j] = row;

    f32x4 h1[HT], h2[HT], out[OT];
    osa_mlp_forward<HT, OT>(nd, p, valid ? a.obs + row * a.ld_obs : nullptr, a.ld_obs, vec_ok, h1, h2,
                            out);
    OSA_TICK(1);
    // ---- loss and dL/d(out), S layout
    f32x4 dO[OT], dLS[OT];
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      dO[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dLS[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (net == 0 && a.loss_kind == 2) {
      // Fisher-vector product (NaturalPG._fvp, natural_pg.py:91-119).  For a Gaussian policy with
      // state-independent log_std the Hessian of mean KL(pi_old || pi_theta) at theta = theta_old is
      // exactly J^T diag(1/sigma^2) J / (M D_a) on the mean-network parameters (plus 2/D_a on
      // log_std, added by osa_fvp_finish): JVP through the network here, the VJP is the ordinary
      // backward pass below with dL/d(out) = (J v) / sigma^2 / (M D_a).  No double backward.
      const float* __restrict__ v = a.vec;
      const float* xrow = valid ? a.obs + row * a.ld_obs : nullptr;
      f32x4 t1[HT], t2[HT], tm[OT];
#pragma unroll
      for (int t = 0; t < HT; ++t) t1[t] = *reinterpret_cast<const f32x4*>(v + nd.ob1 + 16 * t + 4 * g);
      for (int kb = 0; kb < nd.KB; ++kb) {
        const f32x4 x = osa_load_x(xrow, 16 * kb + 4 * g, nd.obs_dim, a.ld_obs, vec_ok);
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(v + nd.oW1 + (long)(16 * t + i) * INP + 16 * kb + 4 * g);
          t1[t] = OSA_MFMA(w.x, x.x, t1[t]);
          t1[t] = OSA_MFMA(w.y, x.y, t1[t]);
          t1[t] = OSA_MFMA(w.z, x.z, t1[t]);
          t1[t] = OSA_MFMA(w.w, x.w, t1[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < HT; ++t) t1[t] = t1[t] * osa_dact4(h1[t], nd.act);
#pragma unroll
      for (int t = 0; t < HT; ++t) t2[t] = *reinterpret_cast<const f32x4*>(v + nd.ob2 + 16 * t + 4 * g);
#pragma unroll
      for (int kb = 0; kb < HT; ++kb) {
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(v + nd.oW2 + (16 * t + i) * H + 16 * kb + 4 * g);
          const f32x4 w = *reinterpret_cast<const f32x4*>(p + nd.oW2 + (16 * t + i) * H + 16 * kb + 4 * g);
          t2[t] = OSA_MFMA(wv.x, h1[kb].x, t2[t]);
          t2[t] = OSA_MFMA(wv.y, h1[kb].y, t2[t]);
          t2[t] = OSA_MFMA(wv.z, h1[kb].z, t2[t]);
          t2[t] = OSA_MFMA(wv.w, h1[kb].w, t2[t]);
          t2[t] = OSA_MFMA(w.x, t1[kb].x, t2[t]);
          t2[t] = OSA_MFMA(w.y, t1[kb].y, t2[t]);
          t2[t] = OSA_MFMA(w.z, t1[kb].z, t2[t]);
          t2[t] = OSA_MFMA(w.w, t1[kb].w, t2[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < HT; ++t) t2[t] = t2[t] * osa_dact4(h2[t], nd.act);
#pragma unroll
      for (int o = 0; o < OT; ++o) tm[o] = *reinterpret_cast<const f32x4*>(v + nd.ob3 + 16 * o + 4 * g);
#pragma unroll
      for (int kb = 0; kb < HT; ++kb) {
#pragma unroll
        for (int o = 0; o < OT; ++o) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(v + nd.oW3 + (16 * o + i) * H + 16 * kb + 4 * g);
          const f32x4 w = *reinterpret_cast<const f32x4*>(p + nd.oW3 + (16 * o + i) * H + 16 * kb + 4 * g);
          tm[o] = OSA_MFMA(wv.x, h2[kb].x, tm[o]);
          tm[o] = OSA_MFMA(wv.y, h2[kb].y, tm[o]);
          tm[o] = OSA_MFMA(wv.z, h2[kb].z, tm[o]);
          tm[o] = OSA_MFMA(wv.w, h2[kb].w, tm[o]);
          tm[o] = OSA_MFMA(w.x, t2[kb].x, tm[o]);
          tm[o] = OSA_MFMA(w.y, t2[kb].y, tm[o]);
          tm[o] = OSA_MFMA(w.z, t2[kb].z, tm[o]);
          tm[o] = OSA_MFMA(w.w, t2[kb].w, tm[o]);
        }
      }
#pragma unroll
      for (int o = 0; o < OT; ++o) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = 16 * o + 4 * g + r;
          if (d < nd.act_dim && valid) {
            const float sd = expf(p[nd.oLS + d]);
            dO[o][r] = tm[o][r] / (sd * sd) * a.fvp_scale;
          }
        }
      }
    } else if (net == 0) {
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
            const float sd = expf(p[nd.oLS + d]);
            const float var = sd * sd;
            const float z = a.act[row * a.ld_act + d] - out[o][r];
            zv[o][r] = z;
            ivar[o][r] = 1.f / var;
            lp += -(z * z) / (2.f * var) - logf(sd) - 0.91893853320467274178f;
          }
        }
      }
      lp = osa_sum_over_groups(lp);
      // ---- optional per-sample KL(pi_theta || pi_old) (torch.distributions.kl._kl_normal_normal)
      const bool ext_kl = a.old_mean != nullptr && (a.ext_kl_coef != 0.f || a.ext_mask_eta >= 0.f);
      float kl = 0.f;
      f32x4 dkl_mu[OT], dkl_ls[OT];
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        dkl_mu[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dkl_ls[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      if (ext_kl) {
#pragma unroll
        for (int o = 0; o < OT; ++o) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int d = 16 * o + 4 * g + r;
            if (d < nd.act_dim && valid) {
              const float sd = expf(p[nd.oLS + d]), sd0 = expf(a.old_log_std[d]);
              const float q = sd / sd0, var_ratio = q * q;
              const float dm = out[o][r] - a.old_mean[row * a.ld_old_mean + d];
              const float u = dm / sd0, t1 = u * u;
              kl += 0.5f * (var_ratio + t1 - 1.f - logf(var_ratio));
              dkl_mu[o][r] = dm / (sd0 * sd0);
              dkl_ls[o][r] = var_ratio - 1.f;
            }
          }
        }
        kl = osa_sum_over_groups(kl);
      }
      const float ratio = valid ? expf(lp - a.logp[row]) : 0.f;
      // ---- P3O: kappa * relu(mean_i(ratio_i * adv_c_i) + (Jc - limit)); the mean needs the whole minibatch
      float cost_w = 0.f;
      if (a.ext_cost_kappa > 0.f) {  // block-uniform; the host guarantees a single chunk
        const float part = (valid && g == 0) ? ratio * a.adv_c[row] : 0.f;
        __syncthreads();
        const float surr_c = osa_block_sum_f(part, red) * invB;
        const float pen = surr_c + a.ext_cost_excess;
        if (pen > 0.f) cost_w = a.ext_cost_kappa;
        if (threadIdx.x == 0) a.stats[10] = a.ext_cost_kappa * fmaxf(pen, 0.f);
        __syncthreads();
      }
      // ---- FOCOPS trust mask.  focops.py:84-88 multiplies a (B,1) KL column, a (B,) ratio*adv row and
      // the (B,1) mask, i.e. the loss it differentiates is the mean of a (B,B) broadcast:
      //     mean_i(mask_i KL_i) - mean_i(mask_i) * mean_j(ratio_j adv_j) / focops_lam
      // -> the KL term is masked per sample, the surrogate term is scaled by the mask's minibatch mean.
      float mask = 1.f, mask_mean = 1.f;
      if (a.ext_mask_eta >= 0.f) {  // block-uniform; single chunk (host-checked)
        mask = (valid && kl <= a.ext_mask_eta) ? 1.f : 0.f;
        __syncthreads();
        mask_mean = osa_block_sum_f((valid && g == 0) ? mask : 0.f, red) * invB;
        __syncthreads();
      }
      if (valid) {
        // PPOLag surrogate advantage (ppo_lag.py:101-102)
        const float adv = (a.adv_r[row] - lam * a.adv_c[row]) / (1.f + lam);
        float dratio, li;
        if (a.loss_kind == 0) {
          const float lo = 1.f - a.hp.clip, hi = 1.f + a.hp.clip;
          const float rc = fminf(fmaxf(ratio, lo), hi);
          const float s1 = ratio * adv, s2 = rc * adv;
          const bool inrange = ratio >= lo && ratio <= hi;
          li = -fminf(s1, s2);
          dratio = (s1 < s2 || inrange) ? -adv : 0.f;
        } else {
          li = -(ratio * adv);
          dratio = -adv;
        }
        const float rs = a.ext_ratio_scale * mask_mean;
        li = li * rs + a.ext_kl_coef * kl * mask;
        dratio = dratio * rs + cost_w * a.adv_c[row];
        const float dlogp = dratio * ratio * invB;
        const float dklw = a.ext_kl_coef * mask * invB;
        if (g == 0) {
          loss_acc += li;
          ratio_acc += ratio;
        }
#pragma unroll
        for (int o = 0; o < OT; ++o) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z = zv[o][r], iv = ivar[o][r];
            // d logp / d mu = z / var ; d logp / d log_std = z^2 / var - 1
            dO[o][r] = dlogp * z * iv + dklw * dkl_mu[o][r];
            dLS[o][r] = (iv != 0.f) ? dlogp * (z * z * iv - 1.f) + dklw * dkl_ls[o][r] : 0.f;
          }
        }
      }
    } else if (valid) {
      const float tgt = (net == 1 ? a.tgt_r : a.tgt_c)[row];
      const float diff = out[0][0] - tgt;
      if (g == 0) {
        loss_acc += diff * diff;
        dO[0][0] = 2.f * diff * invB;
      }
    }
    OSA_TICK(2);
    // ---- backward through the hidden layers (S layout, activations stay in registers)
    const float* __restrict__ W2 = p + nd.oW2;
    const float* __restrict__ W3 = p + nd.oW3;
    f32x4 z2[HT], z1[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int o = 0; o < OT; ++o) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {  // A[i][k] = W3^T[16t+i][16o+4g+s]
          const float w = W3[(16 * o + 4 * g + s) * H + 16 * t + i];
          acc = OSA_MFMA(w, dO[o][s], acc);
        }
      }
      z2[t] = acc * osa_dact4(h2[t], nd.act);  // activation' through its output
    }
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < HT; ++kb) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {  // A[i][k] = W2^T[16t+i][16kb+4g+s]
          const float w = W2[(16 * kb + 4 * g + s) * H + 16 * t + i];
          acc = OSA_MFMA(w, z2[kb][s], acc);
        }
      }
      z1[t] = acc * osa_dact4(h1[t], nd.act);
    }
    OSA_TICK(3);
    // ---- S layout -> F layout through LDS: element (feature f, sample c) at [f*SLD + c]
    const int c = 16 * wave + j;
    if (active) {
#pragma unroll
      for (int t = 0; t < HT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * t + 4 * g + r;
          sH1[f * OSA_SLD + c] = h1[t][r];
          sH2[f * OSA_SLD + c] = h2[t][r];
          sZ1[f * OSA_SLD + c] = z1[t][r];
          sZ2[f * OSA_SLD + c] = z2[t][r];
        }
      }
#pragma unroll
      for (int o = 0; o < OT; ++o) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * o + 4 * g + r;
          sDO[f * OSA_SLD + c] = dO[o][r];
          sDL[f * OSA_SLD + c] = dLS[o][r];
        }
      }
    }
    __syncthreads();
    OSA_TICK(4);
    // ---- weight gradients: contraction over the SPC samples of the chunk.
    // D tile: lane (cc = l&15, g) holds dW[row 4g + r][col cc].  Wave w takes the row tiles w, w + 4, ...
    const int cc = j;
    {
      // this lane's sample rows (constant over the row tiles and K blocks)
      long rows[4 * NSB];
#pragma unroll
      for (int q = 0; q < 4 * NSB; ++q) rows[q] = sIdx[16 * (q >> 2) + 4 * g + (q & 3)];
      for (int rt = wave; rt < HT; rt += 4) {  // dW2 row-tile rt, all HT column tiles;  dW1 row-tile rt, all KB column tiles
        f32x4 a2[NSB], a1[NSB];
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
          a2[sb] = *reinterpret_cast<const f32x4*>(sZ2 + (16 * rt + i) * OSA_SLD + 16 * sb + 4 * g);
          a1[sb] = *reinterpret_cast<const f32x4*>(sZ1 + (16 * rt + i) * OSA_SLD + 16 * sb + 4 * g);
        }
#pragma unroll
        for (int ti = 0; ti < HT; ++ti) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int sb = 0; sb < NSB; ++sb) {
            const f32x4 b =
                *reinterpret_cast<const f32x4*>(sH1 + (16 * ti + i) * OSA_SLD + 16 * sb + 4 * g);
            acc = OSA_MFMA(a2[sb].x, b.x, acc);
            acc = OSA_MFMA(a2[sb].y, b.y, acc);
            acc = OSA_MFMA(a2[sb].z, b.z, acc);
            acc = OSA_MFMA(a2[sb].w, b.w, acc);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* dst = gout + nd.oW2 + (16 * rt + 4 * g + r) * H + 16 * ti + cc;
            *dst = first ? acc[r] : *dst + acc[r];
          }
        }
        // the gathered x values of block kb+1 are requested before the MFMAs of block kb issue
        float xq[4 * NSB], xnq[4 * NSB];
#pragma unroll
        for (int q = 0; q < 4 * NSB; ++q)
          xnq[q] = (rows[q] >= 0 && cc < nd.obs_dim) ? a.obs[rows[q] * a.ld_obs + cc] : 0.f;
        for (int kb = 0; kb < nd.KB; ++kb) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          const int col = 16 * kb + cc;
#pragma unroll
          for (int q = 0; q < 4 * NSB; ++q) xq[q] = xnq[q];
          if (kb + 1 < nd.KB) {
            const int coln = col + 16;
#pragma unroll
            for (int q = 0; q < 4 * NSB; ++q)
              xnq[q] = (rows[q] >= 0 && coln < nd.obs_dim) ? a.obs[rows[q] * a.ld_obs + coln] : 0.f;
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int sb = 0; sb < NSB; ++sb) {
#pragma unroll
            for (int s = 0; s < 4; ++s)  // B[k = sample 16sb+4g+s][j = input feature col]
              acc = OSA_MFMA(a1[sb][s], xq[4 * sb + s], acc);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* dst = gout + nd.oW1 + (long)(16 * rt + 4 * g + r) * INP + col;
            *dst = first ? acc[r] : *dst + acc[r];
          }
        }
      }
    }
    OSA_TICK(5);
    // dW3: output tiles o x column tiles wave, wave + 4, ...
    for (int ct = wave; ct < HT; ct += 4) {
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
          const f32x4 av =
              *reinterpret_cast<const f32x4*>(sDO + (16 * o + i) * OSA_SLD + 16 * sb + 4 * g);
          const f32x4 b =
              *reinterpret_cast<const f32x4*>(sH2 + (16 * ct + i) * OSA_SLD + 16 * sb + 4 * g);
          acc = OSA_MFMA(av.x, b.x, acc);
          acc = OSA_MFMA(av.y, b.y, acc);
          acc = OSA_MFMA(av.z, b.z, acc);
          acc = OSA_MFMA(av.w, b.w, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* dst = gout + nd.oW3 + (16 * o + 4 * g + r) * H + 16 * ct + cc;
          *dst = first ? acc[r] : *dst + acc[r];
        }
      }
    }
    OSA_TICK(6);
    // bias / log_std gradients: one thread per feature sums its LDS row over the 64 samples
    for (int tid = threadIdx.x; tid < 2 * H + 2 * OUTP; tid += blockDim.x) {
      const float* srow;
      float* dst;
      if (tid < H) {
        srow = sZ1 + tid * OSA_SLD;
        dst = gout + nd.ob1 + tid;
      } else if (tid < 2 * H) {
        srow = sZ2 + (tid - H) * OSA_SLD;
        dst = gout + nd.ob2 + (tid - H);
      } else if (tid < 2 * H + OUTP) {
        srow = sDO + (tid - 2 * H) * OSA_SLD;
        dst = gout + nd.ob3 + (tid - 2 * H);
      } else {
        srow = sDL + (tid - 2 * H - OUTP) * OSA_SLD;
        dst = gout + nd.oLS + (tid - 2 * H - OUTP);
      }
      float s = 0.f;
#pragma unroll 16
      for (int k = 0; k < SPC; ++k) s += srow[k];
      *dst = first ? s : *dst + s;
    }
  }  // chunks
  // ---- block-level loss statistics (deterministic order)
  __syncthreads();
  OSA_TICK(7);
  const float loss_sum = osa_block_sum_f(loss_acc, red);
  const float ratio_sum = osa_block_sum_f(ratio_acc, red);
  if (a.nblk > 1) {
    if (threadIdx.x == 0) {
      gout[P + 0] = loss_sum;
      gout[P + 1] = ratio_sum;
    }
    return;
  }
  if (threadIdx.x == 0) {
    if (net == 0) {
      float ent = 0.f;
      for (int d = 0; d < nd.act_dim; ++d) ent += 1.41893853320467274178f + p[nd.oLS + d];
      ent /= (float)nd.act_dim;
      a.stats[2] = loss_sum * invB - a.hp.entropy_coef * ent;
      a.stats[3] = ratio_sum * invB;
      a.stats[4] = ent;
    } else {
      a.stats[net - 1] = loss_sum * invB;
    }
  }
  if (net == 0 && a.hp.entropy_coef != 0.f && threadIdx.x < nd.act_dim)
    gout[nd.oLS + threadIdx.x] -= a.hp.entropy_coef / (float)nd.act_dim;
  __syncthreads();
  OSA_TICK(8);
  osa_finalize_net(a, net, red);
  OSA_TICK(12);
}

// Sum the per-block partial slabs into grads[net] (large-batch path), then finalize in a second launch.
__global__ __launch_bounds__(256) void osa_slab_reduce_kernel(OsaMbArgs a) {
  const OsaNet& nd = a.nd;
  const int net = blockIdx.y;
  if (!((a.nets_mask >> net) & 1)) return;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int W = nd.P + OSA_NSTAT;
  if (e >= W) return;
  const float* s = a.slabs + (long)net * a.nblk * W + e;
  const int ns = a.nslab[0] < 0 ? a.nblk : a.nslab[net];
  // eight independent partial sums (slab b goes to partial b mod 8), combined in a fixed order: eight
  // times the loads in flight of a single running sum, still deterministic
  float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int b = 0;
  for (; b + 8 <= ns; b += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] += s[(long)(b + u) * W];
  }
  for (; b < ns; ++b) p[0] += s[(long)b * W];
  const float acc = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  if (e < nd.P) {
    a.grads[(long)net * nd.P + e] = acc;
  } else {
    const int k = e - nd.P;
    const float invB = 1.f / (float)a.B;
    if (net == 0) {
      if (k == 0 || k == 1) {
        // loss needs the entropy term; log_std is read directly
        if (k == 0) {
          float ent = 0.f;
          const float* p = a.params;
          for (int d = 0; d < nd.act_dim; ++d) ent += 1.41893853320467274178f + p[nd.oLS + d];
          ent /= (float)nd.act_dim;
          a.stats[2] = acc * invB - a.hp.entropy_coef * ent;
          a.stats[4] = ent;
        } else {
          a.stats[3] = acc * invB;
        }
      }
    } else if (k == 0) {
      a.stats[net - 1] = acc * invB;
    }
  }
}

// Large-batch step, second (and last) launch.  Grid (ceil((P+16)/256), 3): every workgroup reduces 256 elements
// of the slabs (coalesced, 8 independent partial sums), forms gradient + L2 term and its share of |g|^2 and
// sum p^2, and publishes the two partial sums; then ALL workgroups of the network meet at an arrival counter
// (release fence -> agent-scope add -> bounded spin -> acquire fence: the ~100 workgroups of a launch are a
// fraction of the chip, co-resident by construction), add the partials in block order (every workgroup gets the
// same total norm, bit for bit) and apply clip + Adam to THEIR OWN 256 parameters.  Round 1 used two more
// launches (slab reduce 7 us, then norm + clip + Adam by one 1024-thread block per network: 12 us).
// sync: int[8] per launch site, zero before the first call; the last workgroup to FINISH resets it.
//   sync[net] arrival counter, sync[4 + net] finish counter, sync[3] sticky time-out flag.
__global__ __launch_bounds__(256) void osa_slab_reduce_finalize_kernel(OsaMbArgs a, float* partials, int* sync) {
  __shared__ float red[32];
  __shared__ float s_tot[2];
  const OsaNet& nd = a.nd;
  const int net = blockIdx.y;
  if (!((a.nets_mask >> net) & 1)) return;
  const int P = nd.P, W = P + OSA_NSTAT, nblk = gridDim.x;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const bool critic = net != 0;
  const int step = a.adam_step[net] + 1;  // read before anybody can advance it (the last finisher does)
  float* __restrict__ p = a.params + (long)net * P;
  float gval = 0.f, pval = 0.f, gsq = 0.f, psq = 0.f;
  float mv0 = 0.f, vv0 = 0.f, step_size = 0.f, inv_bc2_sqrt = 0.f;
  bool pre_done = false;
  if (e < W) {
    const float* s = a.slabs + (long)net * a.nblk * W + e;
    const int ns = a.nslab[0] < 0 ? a.nblk : a.nslab[net];
    float q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // same association as osa_slab_reduce_kernel
    int b = 0;
    // Round 6 (phase clocks of this launch: slab sum 8 700 cycles, then 3 300 for the float64 bias corrections, 3 300 for
    // two block sums, 5 300 barrier, 2 600 for the totals): the first 64 slabs are REQUESTED, then everything of the Adam
    // step that does not depend on them is computed while they travel -- the moments, two pow, a sqrt, two divisions
    if (ns >= 64) {
      float t[64];
#pragma unroll
      for (int u = 0; u < 64; ++u) t[u] = s[(long)u * W];
      __builtin_amdgcn_sched_barrier(0);
      if (e < P && a.mode == 0) {
        mv0 = a.adam_m[(long)net * P + e];
        vv0 = a.adam_v[(long)net * P + e];
        const double b1 = a.hp.beta1, b2 = a.hp.beta2;
        const float lr = a.hp.lr_dev ? a.hp.lr_dev[critic ? 1 : 0] : (critic ? a.hp.lr_critic : a.hp.lr_actor);
        step_size = (float)((double)lr / (1.0 - pow(b1, (double)step)));
        inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow(b2, (double)step)));
        pre_done = true;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < 64; v += 16) {
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] += t[v + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] += t[v + 8 + u];
      }
      b = 64;
    }
    for (; b + 64 <= ns; b += 64) {  // 64 slabs' loads in flight (ONE round trip for the large-batch step's 64
      // slabs instead of four: the loop is latency-bound, 99 workgroups x 256 threads x 4 bytes per load); the
      // additions keep the order of the 16-slab loop below
      float t[64];
#pragma unroll
      for (int u = 0; u < 64; ++u) t[u] = s[(long)(b + u) * W];
#pragma unroll
      for (int v = 0; v < 64; v += 16) {
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] += t[v + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] += t[v + 8 + u];
      }
    }
    for (; b + 16 <= ns; b += 16) {  // 16 slabs' loads in flight; q[u] receives b + u, b + 8 + u in this order
      float t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = s[(long)(b + u) * W];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] += t[u];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] += t[8 + u];
    }
    for (; b + 8 <= ns; b += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] += s[(long)(b + u) * W];
    }
    for (; b < ns; ++b) q[0] += s[(long)b * W];
    const float acc = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
    if (e < P) {
      pval = p[e];
      gval = acc;
      if (net == 0 && a.hp.entropy_coef != 0.f && e >= nd.oLS && e < nd.oLS + nd.act_dim)
        gval -= a.hp.entropy_coef / (float)nd.act_dim;
      if (critic && e >= nd.oLS) {  // critics have no log_std; keep the slot inert
        gval = 0.f;
      } else {
        if (critic && a.hp.use_critic_norm) gval += 2.f * a.hp.critic_norm_coef * pval;
        if (critic) psq = pval * pval;
      }
      gsq = gval * gval;
    } else {
      const int k = e - P;
      const float invB = 1.f / (float)a.B;
      if (net == 0) {
        if (k == 0) {
          float ent = 0.f;
          for (int d = 0; d < nd.act_dim; ++d) ent += 1.41893853320467274178f + a.params[nd.oLS + d];
          ent /= (float)nd.act_dim;
          a.stats[2] = acc * invB - a.hp.entropy_coef * ent;
          a.stats[4] = ent;
        } else if (k == 1) {
          a.stats[3] = acc * invB;
        }
      } else if (k == 0) {
        a.stats[net - 1] = acc * invB;
      }
    }
  }
  // everything of the Adam step that does not depend on the clip factor is requested / computed BEFORE the grid barrier
  // (round 4): the moments of this thread's element and the float64 bias corrections (two pow, a sqrt and two divisions
  // per thread) used to sit behind it, on the critical path of every large-batch step
  if (e < P && a.mode == 0 && !pre_done) {
    mv0 = a.adam_m[(long)net * P + e];
    vv0 = a.adam_v[(long)net * P + e];
    const double b1 = a.hp.beta1, b2 = a.hp.beta2;
    const float lr = a.hp.lr_dev ? a.hp.lr_dev[critic ? 1 : 0] : (critic ? a.hp.lr_critic : a.hp.lr_actor);
    step_size = (float)((double)lr / (1.0 - pow(b1, (double)step)));
    inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow(b2, (double)step)));
  }
  // (one pass for both sums: DPP wave sums, one barrier -- osa_block_sum_f twice was six barriers and twelve
  // ds_bpermute rounds)
  {
    const float wg_ = osa_wave_sum_dpp(gsq), wp_ = osa_wave_sum_dpp(psq);
    if ((threadIdx.x & 63) == 0) {
      red[threadIdx.x >> 6] = wg_;
      red[4 + (threadIdx.x >> 6)] = wp_;
    }
    __syncthreads();
    gsq = (red[0] + red[1]) + (red[2] + red[3]);
    psq = (red[4] + red[5]) + (red[6] + red[7]);
  }
  float* part = partials + ((long)net * nblk + blockIdx.x) * 2;
  if (threadIdx.x == 0) {
    // (agent-scope accesses for the two partial sums instead of an agent-scope release / acquire fence pair around
    // the barrier: the release fence writes back every dirty line of the XCC's L2)
    __hip_atomic_store(part, gsq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(part + 1, psq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // ---- grid barrier of this network's workgroups
  __syncthreads();
  if (threadIdx.x == 0) {
    int seen = __hip_atomic_fetch_add(sync + net, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    int spins = 0;
    while (seen < nblk) {
      __builtin_amdgcn_s_sleep(1);
      seen = __hip_atomic_load(sync + net, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (++spins > (1 << 22)) {  // never hang the device: flag it (the caller checks sync[3])
        __hip_atomic_store(sync + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
  {
    // every partial is fetched by its own thread (one round trip for all); tiles of 256 partials summed by DPP wave sums
    // and a fixed cross-wave order, tile after tile: identical totals in every workgroup (round 6: thread 0 used to add
    // them one by one out of LDS, 2 600 cycles behind the barrier)
    __shared__ float s_w[2][4];
    float* pp = partials + (long)net * nblk * 2;
    float tg = 0.f, tp = 0.f;
    for (int k0 = 0; k0 < nblk; k0 += 256) {
      const int k = k0 + threadIdx.x;
      float pg = 0.f, ppv = 0.f;
      if (k < nblk) {
        pg = __hip_atomic_load(pp + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ppv = __hip_atomic_load(pp + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      pg = osa_wave_sum_dpp(pg);
      ppv = osa_wave_sum_dpp(ppv);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) {
        s_w[0][threadIdx.x >> 6] = pg;
        s_w[1][threadIdx.x >> 6] = ppv;
      }
      __syncthreads();
      tg += (s_w[0][0] + s_w[0][1]) + (s_w[0][2] + s_w[0][3]);
      tp += (s_w[1][0] + s_w[1][1]) + (s_w[1][2] + s_w[1][3]);
    }
    if (threadIdx.x == 0) {
      s_tot[0] = tg;
      s_tot[1] = tp;
    }
  }
  __syncthreads();
  const float total_norm = sqrtf(s_tot[0]);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.stats[7 + net] = total_norm;
    if (critic) a.stats[4 + net] = s_tot[1];
  }
  float coef = 1.f;
  if (a.hp.use_max_grad_norm && a.mode != 2) {
    coef = a.hp.max_grad_norm / (total_norm + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
  }
  if (e < P) {
    if (a.mode != 0) {  // 1: the locally clipped gradient is the result (all-reduce follows); 2: raw gradient
      a.grads[(long)net * P + e] = gval * coef;
    } else {
      float* __restrict__ m = a.adam_m + (long)net * P;
      float* __restrict__ v = a.adam_v + (long)net * P;
      float mv = mv0, vv = vv0;
      p[e] = osa_adam_update(gval * coef, mv, vv, pval, a.hp.beta1, a.hp.beta2, step_size, inv_bc2_sqrt,
                             a.hp.adam_eps);
      m[e] = mv;
      v[e] = vv;
    }
  }
  // ---- the last workgroup to finish advances the step counter and re-arms the counters
  __syncthreads();
  if (threadIdx.x == 0) {
    const int done = __hip_atomic_fetch_add(sync + 4 + net, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    if (done == nblk) {
      if (a.mode == 0) a.adam_step[net] = step;
      __hip_atomic_store(sync + net, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sync + 4 + net, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ __launch_bounds__(1024) void osa_finalize_kernel(OsaMbArgs a, int add_entropy_grad) {
  __shared__ float red[32];
  const int net = blockIdx.y;
  if (!((a.nets_mask >> net) & 1)) return;
  if (add_entropy_grad && net == 0 && a.hp.entropy_coef != 0.f && threadIdx.x < a.nd.act_dim)
    a.grads[a.nd.oLS + threadIdx.x] -= a.hp.entropy_coef / (float)a.nd.act_dim;
  __syncthreads();
  OSA_TICK(8);
  osa_finalize_net(a, net, red);
  OSA_TICK(12);
}

// ------------------------------------------------------------------------------------------------
// K11  full-batch KL(old || new), policy_gradient.py:383-389: kl_divergence(..).sum(-1).mean()
//      (reduce_mode 0) or .mean() over all M x D_a elements (reduce_mode 1: natural_pg.py:95,
//      trpo.py:113, cpo.py:133).  Also snapshots the mean (old distribution) when mean_out != null.
// ------------------------------------------------------------------------------------------------
template <int HT, int OT>
__global__ __launch_bounds__(256) void osa_actor_kl_kernel(
    OsaNet nd, const float* __restrict__ params, const float* __restrict__ obs, int ld, long M,
    const float* __restrict__ old_mean, int ld_old, const float* __restrict__ old_log_std,
    float* __restrict__ mean_out, int ld_mean, double* __restrict__ ws) {
  __shared__ double red[17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const bool vec_ok = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(obs) & 15) == 0);
  double acc = 0.0;
  for (long base = (long)blockIdx.x * 64; base < M; base += (long)gridDim.x * 64) {
    const long row = base + 16 * wave + j;
    const bool valid = row < M;
    f32x4 h1[HT], h2[HT], out[OT];
    osa_mlp_forward<HT, OT>(nd, params, valid ? obs + row * ld : nullptr, ld, vec_ok, h1, h2, out);
#pragma unroll
    for (int o = 0; o < OT; ++o) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = 16 * o + 4 * g + r;
        if (d < nd.act_dim && valid) {
          const float mu = out[o][r];
          if (mean_out) mean_out[row * ld_mean + d] = mu;
          if (old_mean) {
            // torch _kl_normal_normal: 0.5 * (var_ratio + t1 - 1 - log(var_ratio))
            const float ps = expf(old_log_std[d]), qs = expf(params[nd.oLS + d]);
            const float vr = (ps / qs) * (ps / qs);
            const float t1 = ((old_mean[row * ld_old + d] - mu) / qs);
            acc += (double)(0.5f * (vr + t1 * t1 - 1.f - logf(vr)));
          }
        }
      }
    }
  }
  acc = osa_block_sum<256>(acc, red);
  if (threadIdx.x == 0 && ws) ws[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void osa_kl_final_kernel(const double* __restrict__ ws, int nblk,
                                                           double denom, float* __restrict__ out) {
  __shared__ double red[17];
  double s = 0.0;
  for (int k = threadIdx.x; k < nblk; k += 256) s += ws[k];
  s = osa_block_sum<256>(s, red);
  if (threadIdx.x == 0) *out = (float)(s / denom);
}

// ------------------------------------------------------------------------------------------------
// K14  full-batch evaluation of a candidate actor (line searches of TRPO._search_step_size
//      trpo.py:93-148 and CPO._cpo_search_step cpo.py:106-180): sums of ratio*adv (surrogate),
//      ratio*adv_c, KL(old || new) and ratio over all rows.
// ------------------------------------------------------------------------------------------------
template <int HT, int OT>
__global__ __launch_bounds__(256) void osa_actor_eval_kernel(
    OsaNet nd, const float* __restrict__ params, const float* __restrict__ obs, int ld, long M,
    const float* __restrict__ act, int ld_act, const float* __restrict__ logp,
    const float* __restrict__ adv_r, const float* __restrict__ adv_c,
    const float* __restrict__ lagrange, const float* __restrict__ old_mean, int ld_old,
    const float* __restrict__ old_log_std, double* __restrict__ ws) {
  __shared__ double red[17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const bool vec_ok = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(obs) & 15) == 0);
  const float lam = lagrange ? *lagrange : 0.f;
  double s_sur = 0.0, s_cost = 0.0, s_kl = 0.0, s_ratio = 0.0;
  for (long base = (long)blockIdx.x * 64; base < M; base += (long)gridDim.x * 64) {
    const long row = base + 16 * wave + j;
    const bool valid = row < M;
    f32x4 h1[HT], h2[HT], out[OT];
    osa_mlp_forward<HT, OT>(nd, params, valid ? obs + row * ld : nullptr, ld, vec_ok, h1, h2, out);
    float lp = 0.f, klp = 0.f;
#pragma unroll
    for (int o = 0; o < OT; ++o) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = 16 * o + 4 * g + r;
        if (d < nd.act_dim && valid) {
          const float mu = out[o][r];
          const float qs = expf(params[nd.oLS + d]);
          const float z = act[row * ld_act + d] - mu;
          lp += -(z * z) / (2.f * (qs * qs)) - logf(qs) - 0.91893853320467274178f;
          const float ps = expf(old_log_std[d]);
          const float vr = (ps / qs) * (ps / qs);
          const float t1 = (old_mean[row * ld_old + d] - mu) / qs;
          klp += 0.5f * (vr + t1 * t1 - 1.f - logf(vr));
        }
      }
    }
    lp = osa_sum_over_groups(lp);
    klp = osa_sum_over_groups(klp);
    if (valid && g == 0) {
      const float ratio = expf(lp - logp[row]);
      const float adv = (adv_r[row] - lam * adv_c[row]) / (1.f + lam);
      s_sur += (double)(ratio * adv);
      s_cost += (double)(ratio * adv_c[row]);
      s_kl += (double)klp;
      s_ratio += (double)ratio;
    }
  }
  s_sur = osa_block_sum<256>(s_sur, red);
  s_cost = osa_block_sum<256>(s_cost, red);
  s_kl = osa_block_sum<256>(s_kl, red);
  s_ratio = osa_block_sum<256>(s_ratio, red);
  if (threadIdx.x == 0) {
    ws[4 * blockIdx.x + 0] = s_sur;
    ws[4 * blockIdx.x + 1] = s_cost;
    ws[4 * blockIdx.x + 2] = s_kl;
    ws[4 * blockIdx.x + 3] = s_ratio;
  }
}

__global__ __launch_bounds__(256) void osa_eval_final_kernel(const double* __restrict__ ws, int nblk,
                                                             double M, double act_dim,
                                                             float* __restrict__ out4) {
  __shared__ double red[17];
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (int k = threadIdx.x; k < nblk; k += 256)
    for (int q = 0; q < 4; ++q) s[q] += ws[4 * k + q];
  for (int q = 0; q < 4; ++q) s[q] = osa_block_sum<256>(s[q], red);
  if (threadIdx.x == 0) {
    out4[0] = (float)(-s[0] / M);            // loss_pi = -mean(ratio * adv)   (policy_gradient.py:574-578)
    out4[1] = (float)(s[1] / M);             // loss_cost = mean(ratio * adv_c) (cpo.py:209-212)
    out4[2] = (float)(s[2] / (M * act_dim)); // kl_divergence(p, q).mean()      (trpo.py:113, cpo.py:133)
    out4[3] = (float)(s[3] / M);
  }
}

// ------------------------------------------------------------------------------------------------
// K13/K16  flat-vector algebra of conjugate gradients (omnisafe/utils/math.py:116-132) on padded
// parameter vectors (padding entries are zero in every vector, so dot products are unaffected).
// One workgroup; scalars stay on the device: scal[0] = r.r, scal[1] = done flag, scal[2] = p.z.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double osa_dot_block(const float* __restrict__ x, const float* __restrict__ y,
                                                int n, double* red) {
  double s = 0.0;
  for (int e = threadIdx.x; e < n; e += blockDim.x) s += (double)x[e] * (double)y[e];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  s = osa_wave_sum(s);
  __syncthreads();
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < nw; ++w) t += red[w];
    red[16] = t;
  }
  __syncthreads();
  return red[16];
}

__global__ __launch_bounds__(1024) void osa_cg_init_kernel(int n, const float* __restrict__ b,
                                                           float* __restrict__ x, float* __restrict__ r,
                                                           float* __restrict__ p, float* __restrict__ scal) {
  __shared__ double red[17];
  for (int e = threadIdx.x; e < n; e += blockDim.x) {  // x = 0; r = b - F(0) = b; p = r
    x[e] = 0.f;
    r[e] = b[e];
    p[e] = b[e];
  }
  const double rr = osa_dot_block(b, b, n, red);
  if (threadIdx.x == 0) {
    scal[0] = (float)rr;
    scal[1] = 0.f;
  }
}

__global__ __launch_bounds__(1024) void osa_cg_step_kernel(int n, const float* __restrict__ z,
                                                           float* __restrict__ x, float* __restrict__ r,
                                                           float* __restrict__ p, float* __restrict__ scal,
                                                           float residual_tol, float eps) {
  __shared__ double red[17];
  if (scal[1] != 0.f) return;  // the reference broke out of the loop: x is final
  const float rdotr = scal[0];
  const float pz = (float)osa_dot_block(p, z, n, red);
  const float alpha = rdotr / (pz + eps);
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    x[e] += alpha * p[e];
    r[e] -= alpha * z[e];
  }
  __syncthreads();
  const float new_rdotr = (float)osa_dot_block(r, r, n, red);
  if (sqrtf(new_rdotr) < residual_tol) {
    if (threadIdx.x == 0) scal[1] = 1.f;
    return;
  }
  const float mu = new_rdotr / (rdotr + eps);
  for (int e = threadIdx.x; e < n; e += blockDim.x) p[e] = r[e] + mu * p[e];
  if (threadIdx.x == 0) scal[0] = new_rdotr;
}

// out = raw + damping * v, plus the analytic log_std block of the Fisher matrix (2 / D_a) * v[ls]
__global__ __launch_bounds__(1024) void osa_fvp_finish_kernel(int n, const float* __restrict__ raw,
                                                              const float* __restrict__ v, float damping,
                                                              int ls_off, int ls_n, float ls_coef,
                                                              float* __restrict__ out) {
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    float o = raw[e] + damping * v[e];
    if (e >= ls_off && e < ls_off + ls_n) o += ls_coef * v[e];
    out[e] = o;
  }
}

__global__ __launch_bounds__(1024) void osa_vec_lincomb_kernel(int n, float a, const float* __restrict__ x,
                                                               float b, const float* __restrict__ y,
                                                               float* __restrict__ out) {
  for (int e = threadIdx.x + blockIdx.x * blockDim.x; e < n; e += blockDim.x * gridDim.x)
    out[e] = a * x[e] + (y ? b * y[e] : 0.f);
}

__global__ __launch_bounds__(1024) void osa_vec_dot_kernel(int n, const float* __restrict__ x,
                                                           const float* __restrict__ y,
                                                           float* __restrict__ out) {
  __shared__ double red[17];
  const double s = osa_dot_block(x, y, n, red);
  if (threadIdx.x == 0) *out = (float)s;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
// samples per chunk of osa_mb_grad_kernel: 64 while its four [H][SPC + 4] tiles fit the LDS, 32 at width 256
static int osa_mb_spc(const OsaNet& nd) { return nd.H > 128 ? 32 : 64; }
static size_t osa_mb_lds_bytes(const OsaNet& nd) {
  const int spc = osa_mb_spc(nd);
  return (size_t)(4 * nd.H + 2 * nd.OUTP) * (spc + 4) * sizeof(float) + 32 * sizeof(float) + (size_t)spc * sizeof(long);
}

static int osa_check_dims(int obs_dim, int act_dim, int hidden) {
  if (obs_dim < 1 || act_dim < 1) return OSA_EINVAL;
  // hidden_sizes [H, H]: 64 (all BASELINE configs; the persistent passes), and 32 / 128 / 256 on this per-step family
  const int H_ = hidden & 0xFFFF;
  if (H_ != 32 && H_ != 64 && H_ != 128 && H_ != 256) return OSA_EUNSUPPORTED;
  if ((hidden >> 16) > OSA_ACT_IDENTITY) return OSA_EUNSUPPORTED;  // activation code (mlp_device.h)
  if (act_dim > 32) return OSA_EUNSUPPORTED;
  return OSA_OK;
}

// CALL(HT, OT, NSB): hidden tiles, output tiles, 16-sample blocks per chunk of osa_mb_grad_kernel (ignored by the
// forward-only kernels)
#define OSA_DISPATCH_OT(nd, CALL)                                   \
  do {                                                              \
    const int ot_ = (nd).OUTP == 16 ? 1 : 2;                        \
    if ((nd).H == 64) { if (ot_ == 1) { CALL(4, 1, 4); } else { CALL(4, 2, 4); } }          \
    else if ((nd).H == 128) { if (ot_ == 1) { CALL(8, 1, 4); } else { CALL(8, 2, 4); } }    \
    else if ((nd).H == 256) { if (ot_ == 1) { CALL(16, 1, 2); } else { CALL(16, 2, 2); } }  \
    else { if (ot_ == 1) { CALL(2, 1, 4); } else { CALL(2, 2, 4); } }                       \
  } while (0)

static long long* g_osa_dbg_clocks = nullptr;

// fvp_kernel.hip: the Fisher-vector product with the parameters in LDS (OSA_EUNSUPPORTED: shape outside that kernel)
int osa_launch_fvp_fast(const OsaNet& nd, const float* params, float* grads, const float* obs, int ld_obs, long M,
                        const float* vec, int max_blocks, float* ws, float* step_stats, hipStream_t st);

extern "C" {

int osa_debug_set_clock_buffer(long long* dev_ptr) {
  g_osa_dbg_clocks = dev_ptr;
  return OSA_OK;
}

int osa_mlp_layout(int obs_dim, int act_dim, int hidden, int* out12) {
  OSA_REQUIRE(out12 != nullptr);
  const int rc = osa_check_dims(obs_dim, act_dim, hidden);
  if (rc != OSA_OK) return rc;
  const OsaNet n = osa_make_net(obs_dim, act_dim, hidden);
  const int v[12] = {n.INP, n.OUTP, n.oW1, n.ob1, n.oW2, n.ob2, n.oW3, n.ob3, n.oLS, n.P, n.H, n.KB};
  for (int k = 0; k < 12; ++k) out12[k] = v[k];
  return OSA_OK;
}

int osa_policy_step(int obs_dim, int act_dim, int hidden, const float* params, const float* obs,
                    int ld_obs, int N, const float* eps, unsigned long long seed,
                    unsigned long long offset, const unsigned long long* offset_base, int deterministic,
                    int nets_mask, float* act, int ld_act, float* value_r, float* value_c, float* logp,
                    float* mean_out, int ld_mean, void* stream) {
  return osa_policy_step_scaled(obs_dim, act_dim, hidden, params, obs, ld_obs, N, eps, seed, offset, offset_base,
                                deterministic, nets_mask, act, ld_act, value_r, value_c, logp, mean_out, ld_mean,
                                nullptr, 0, nullptr, nullptr, 0.f, 1.f, stream);
}

int osa_policy_step_scaled(int obs_dim, int act_dim, int hidden, const float* params, const float* obs,
                           int ld_obs, int N, const float* eps, unsigned long long seed,
                           unsigned long long offset, const unsigned long long* offset_base, int deterministic,
                           int nets_mask, float* act, int ld_act, float* value_r, float* value_c, float* logp,
                           float* mean_out, int ld_mean, float* act_env, int ld_env, const float* old_min,
                           const float* old_max, float min_action, float max_action, void* stream) {
  const int rc = osa_check_dims(obs_dim, act_dim, hidden);
  if (rc != OSA_OK) return rc;
  OSA_REQUIRE(params && obs && N > 0 && ld_obs >= obs_dim);
  OSA_REQUIRE(!act || ld_act >= act_dim);
  OSA_REQUIRE(!act_env || (ld_env >= act_dim && old_min && old_max && max_action != min_action));
  const OsaNet nd = osa_make_net(obs_dim, act_dim, hidden);
  const dim3 grid((N + 63) / 64, 3);
#define OSA_CALL(HT, OT, NSB)                                                                          \
  hipLaunchKernelGGL((osa_policy_step_kernel<HT, OT>), grid, dim3(256), 0, osa_stream(stream), nd, \
                     params, obs, ld_obs, N, eps, seed, offset, offset_base, deterministic, nets_mask, act, \
                     ld_act, value_r, value_c, logp, mean_out, ld_mean, act_env, ld_env, old_min, old_max,    \
                     min_action, max_action)
  OSA_DISPATCH_OT(nd, OSA_CALL);
#undef OSA_CALL
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

size_t osa_minibatch_ws_floats(int obs_dim, int act_dim, int hidden, int max_blocks) {
  if (osa_check_dims(obs_dim, act_dim, hidden) != OSA_OK || max_blocks < 1) return 0;
  const OsaNet nd = osa_make_net(obs_dim, act_dim, hidden);
  // slabs + the fused reduce/finalize launch's partial norms [3][ceil((P+16)/256)][2] + its counters (int[8])
  return (size_t)3 * max_blocks * (nd.P + OSA_NSTAT) + (size_t)3 * ((nd.P + OSA_NSTAT + 255) / 256) * 2 + 8;
}

int osa_ppo_minibatch(int obs_dim, int act_dim, int hidden, float* params, float* adam_m,
                      float* adam_v, int* adam_step, float* grads, const float* obs, int ld_obs,
                      const float* act, int ld_act, const float* logp, const float* target_value_r,
                      const float* target_value_c, const float* adv_r, const float* adv_c,
                      const long* idx, int B, const float* lagrange, const osa_ppo_hparams* hp,
                      int loss_kind, int mode, int nets_mask, int max_blocks, float* ws,
                      float* step_stats, void* stream) {
  return osa_ppo_minibatch_ext(obs_dim, act_dim, hidden, params, adam_m, adam_v, adam_step, grads, obs,
                               ld_obs, act, ld_act, logp, target_value_r, target_value_c, adv_r, adv_c,
                               idx, B, lagrange, hp, loss_kind, mode, nets_mask, max_blocks, ws,
                               step_stats, nullptr, stream);
}

int osa_ppo_minibatch_ext(int obs_dim, int act_dim, int hidden, float* params, float* adam_m,
                          float* adam_v, int* adam_step, float* grads, const float* obs, int ld_obs,
                          const float* act, int ld_act, const float* logp,
                          const float* target_value_r, const float* target_value_c,
                          const float* adv_r, const float* adv_c, const long* idx, int B,
                          const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int mode,
                          int nets_mask, int max_blocks, float* ws, float* step_stats,
                          const osa_surrogate_ext* ext, void* stream) {
  const int rc = osa_check_dims(obs_dim, act_dim, hidden);
  if (rc != OSA_OK) return rc;
  OSA_REQUIRE(params && adam_m && adam_v && adam_step && grads && obs && act && logp && hp);
  OSA_REQUIRE(target_value_r && target_value_c && adv_r && adv_c && step_stats && B > 0);
  OSA_REQUIRE(ld_obs >= obs_dim && ld_act >= act_dim && mode >= 0 && mode <= 2);
  OsaMbArgs a = {};
  a.nslab[0] = -1;
  a.ext_ratio_scale = 1.f; a.ext_mask_eta = -1.f;
  if (ext) {
    const bool need_old = ext->kl_coef != 0.f || ext->kl_mask_eta >= 0.f;
    OSA_REQUIRE(!need_old || (ext->old_mean && ext->old_log_std && ext->ld_old_mean >= act_dim));
    // P3O's penalty and FOCOPS' mask mean are minibatch-level quantities: one 64-row block only
    if ((ext->cost_kappa > 0.f || ext->kl_mask_eta >= 0.f) && B > 64) return OSA_EUNSUPPORTED;
    a.old_mean = need_old ? ext->old_mean : nullptr; a.ld_old_mean = ext->ld_old_mean;
    a.old_log_std = ext->old_log_std; a.ext_kl_coef = ext->kl_coef; a.ext_mask_eta = ext->kl_mask_eta;
    a.ext_ratio_scale = ext->ratio_scale; a.ext_cost_kappa = ext->cost_kappa;
    a.ext_cost_excess = ext->cost_excess;
  }
  a.nd = osa_make_net(obs_dim, act_dim, hidden);
  a.params = params; a.adam_m = adam_m; a.adam_v = adam_v; a.adam_step = adam_step; a.grads = grads;
  a.obs = obs; a.ld_obs = ld_obs; a.act = act; a.ld_act = ld_act; a.logp = logp;
  a.tgt_r = target_value_r; a.tgt_c = target_value_c; a.adv_r = adv_r; a.adv_c = adv_c;
  a.idx = idx; a.B = B; a.lagrange = lagrange;
  a.hp.clip = hp->clip; a.hp.entropy_coef = hp->entropy_coef;
  a.hp.critic_norm_coef = hp->critic_norm_coef; a.hp.max_grad_norm = hp->max_grad_norm;
  a.hp.lr_actor = hp->lr_actor; a.hp.lr_critic = hp->lr_critic; a.hp.lr_dev = hp->lr_device; a.hp.beta1 = hp->beta1;
  a.hp.beta2 = hp->beta2; a.hp.adam_eps = hp->adam_eps; a.hp.use_critic_norm = hp->use_critic_norm;
  a.hp.use_max_grad_norm = hp->use_max_grad_norm; a.hp.use_cost = hp->use_cost;
  a.mode = mode; a.stats = step_stats; a.loss_kind = loss_kind;
  a.nets_mask = nets_mask & (hp->use_cost ? 7 : 3);
  a.dbg = g_osa_dbg_clocks;
  a.vec = nullptr; a.fvp_scale = 0.f;
  const int spc = osa_mb_spc(a.nd);
  // (the minibatch-level quantities of P3O / FOCOPS need the whole minibatch in ONE chunk: 32 rows at width 256)
  if (ext && (ext->cost_kappa > 0.f || ext->kl_mask_eta >= 0.f) && B > spc) return OSA_EUNSUPPORTED;
  const int nchunk = (B + spc - 1) / spc;
  int nblk = nchunk;
  if (max_blocks < 1) max_blocks = 1;
  if (nblk > max_blocks) nblk = max_blocks;
  if (nblk > 1) OSA_REQUIRE(ws != nullptr);
  a.nblk = nblk; a.slabs = ws;
  // partial norms and counters of the fused slab reduce + clip/Adam launch live behind the slabs
  const int rblk = (a.nd.P + OSA_NSTAT + 255) / 256;
  float* partials = ws ? ws + (size_t)3 * max_blocks * (a.nd.P + OSA_NSTAT) : nullptr;
  int* tickets = ws ? reinterpret_cast<int*>(partials + (size_t)3 * rblk * 2) : nullptr;
  // Large minibatches: the gradient on the persistent kernel's machinery -- min(nblk, ~CUs/3) workgroups per
  // network keep the weights in LDS and their partial gradient in registers over several 64-row chunks and
  // write ONE slab each (this kernel re-reads the weights from L2 for every chunk and read-modify-writes
  // its slab in global memory per chunk).  Same slab reduce + clip/Adam afterwards.
  if (!ext && B >= 2048 && nblk > 1 && loss_kind <= 1) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= 3) {
      // Round 4: BALANCED partial gradients -- the 64-row chunk-tasks of all networks of the mask laid end to end
      // and shared evenly by one workgroup per compute unit (16 384 rows x 3 networks = 768 tasks = 3 per unit;
      // the strided form below runs 64 workgroups x 4 chunks per network on 192 units).  OSA_LARGE_BATCH_BALANCED=0
      // keeps the strided form (A/B switch).
      const char* bal_env = getenv("OSA_LARGE_BATCH_BALANCED");  // (read per call: the tests toggle it in-process)
      const bool balanced = !(bal_env && bal_env[0] == '0');
      if (balanced) {
        int stride = 0;
        const int brc = osa_pass_partial_grad_balanced(obs_dim, act_dim, hidden, params, obs, ld_obs, act, ld_act, logp,
                                                       target_value_r, target_value_c, adv_r, adv_c, idx, B, lagrange,
                                                       hp, loss_kind, a.nets_mask, cus, max_blocks, ws, a.nslab, &stride,
                                                       stream);
        if (brc == OSA_OK) {
          a.nblk = stride;
          const int W = a.nd.P + OSA_NSTAT;
          hipLaunchKernelGGL(osa_slab_reduce_finalize_kernel, dim3((W + 255) / 256, 3), dim3(256), 0,
                             osa_stream(stream), a, partials, tickets);
          OSA_CHECK_LAUNCH();
          return OSA_OK;
        }
        a.nslab[0] = -1;
        if (brc != OSA_EUNSUPPORTED) return brc;
      }
      int pb = cus / 3;
      if (pb > nblk) pb = nblk;
      // the slowest workgroup walks through ceil(nchunk / pb) chunks whatever happens: take the FEWEST workgroups
      // with that maximum (16 384 rows = 256 chunks: 64 workgroups x 4 chunks instead of 85 of which one has 4) --
      // fewer slabs for the reduction that follows, the same critical path
      {
        const int per = (nchunk + pb - 1) / pb;
        pb = (nchunk + per - 1) / per;
      }
      const int prc = osa_pass_partial_grad(obs_dim, act_dim, hidden, params, obs, ld_obs, act, ld_act, logp,
                                            target_value_r, target_value_c, adv_r, adv_c, idx, B, lagrange,
                                            hp, loss_kind, a.nets_mask, pb, ws, stream);
      if (prc == OSA_OK) {
        a.nblk = pb;
        const int W = a.nd.P + OSA_NSTAT;
        hipLaunchKernelGGL(osa_slab_reduce_finalize_kernel, dim3((W + 255) / 256, 3), dim3(256), 0,
                           osa_stream(stream), a, partials, tickets);
        OSA_CHECK_LAUNCH();
        return OSA_OK;
      }
      if (prc != OSA_EUNSUPPORTED) return prc;
    }
  }
  const size_t lds = osa_mb_lds_bytes(a.nd);
#define OSA_CALL(HT, OT, NSB)                                                                          \
  do {                                                                                            \
    static OsaPerDeviceOnce attr_set;                                                                 \
    if (attr_set.need()) {                                                                              \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&osa_mb_grad_kernel<HT, OT, NSB>),         \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) !=    \
          hipSuccess)                                                                             \
        return OSA_EHIP;                                                                          \
      attr_set.set();                                                                            \
    }                                                                                             \
    hipLaunchKernelGGL((osa_mb_grad_kernel<HT, OT, NSB>), dim3(nblk, 3), dim3(256), lds,               \
                       osa_stream(stream), a);                                                    \
  } while (0)
  OSA_DISPATCH_OT(a.nd, OSA_CALL);
#undef OSA_CALL
  if (nblk > 1) {
    const int W = a.nd.P + OSA_NSTAT;
    hipLaunchKernelGGL(osa_slab_reduce_finalize_kernel, dim3((W + 255) / 256, 3), dim3(256), 0,
                       osa_stream(stream), a, partials, tickets);
  }
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_adam_apply(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                   int* adam_step, float* grads, const osa_ppo_hparams* hp, int nets_mask,
                   void* stream) {
  const int rc = osa_check_dims(obs_dim, act_dim, hidden);
  if (rc != OSA_OK) return rc;
  OSA_REQUIRE(params && adam_m && adam_v && adam_step && grads && hp);
  OsaMbArgs a = {};
  a.nslab[0] = -1;
  a.nd = osa_make_net(obs_dim, act_dim, hidden);
  a.params = params; a.adam_m = adam_m; a.adam_v = adam_v; a.adam_step = adam_step; a.grads = grads;
  a.hp.lr_actor = hp->lr_actor; a.hp.lr_critic = hp->lr_critic; a.hp.lr_dev = hp->lr_device; a.hp.beta1 = hp->beta1;
  a.hp.beta2 = hp->beta2; a.hp.adam_eps = hp->adam_eps;
  a.nets_mask = nets_mask;
  a.mode = 3;
  hipLaunchKernelGGL(osa_finalize_kernel, dim3(1, 3), dim3(1024), 0, osa_stream(stream), a, 0);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_actor_fvp_raw(int obs_dim, int act_dim, int hidden, float* params, float* grads,
                      const float* obs, int ld_obs, long M, const float* vec, int max_blocks,
                      float* ws, float* step_stats, void* stream) {
  const int rc = osa_check_dims(obs_dim, act_dim, hidden);
  if (rc != OSA_OK) return rc;
  OSA_REQUIRE(params && grads && obs && vec && ws && step_stats && M > 0 && ld_obs >= obs_dim);
  {  // hidden width 64, observations up to 64 wide: both parameter blocks in LDS, gradient in registers (fvp_kernel.hip)
    const int rc2 = osa_launch_fvp_fast(osa_make_net(obs_dim, act_dim, hidden), params, grads, obs, ld_obs, M, vec,
                                        max_blocks, ws, step_stats, osa_stream(stream));
    if (rc2 != OSA_EUNSUPPORTED) return rc2;
  }
  OsaMbArgs a = {};
  a.nslab[0] = -1;
  a.nd = osa_make_net(obs_dim, act_dim, hidden);
  a.params = params; a.grads = grads; a.obs = obs; a.ld_obs = ld_obs;
  a.B = (int)M; a.idx = nullptr; a.mode = 2; a.stats = step_stats; a.loss_kind = 2; a.nets_mask = 1;
  a.vec = vec; a.fvp_scale = (float)(1.0 / ((double)M * act_dim));
  a.ext_ratio_scale = 1.f; a.ext_mask_eta = -1.f;
  a.hp.use_cost = 1;
  const int spc = osa_mb_spc(a.nd);
  const int nchunk = (int)((M + spc - 1) / spc);
  int nblk = nchunk;
  if (max_blocks < 1) max_blocks = 1;
  if (nblk > max_blocks) nblk = max_blocks;
  a.nblk = nblk; a.slabs = ws;
  const size_t lds = osa_mb_lds_bytes(a.nd);
#define OSA_CALL(HT, OT, NSB)                                                                          \
  do {                                                                                            \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&osa_mb_grad_kernel<HT, OT, NSB>),           \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) !=      \
        hipSuccess)                                                                               \
      return OSA_EHIP;                                                                            \
    hipLaunchKernelGGL((osa_mb_grad_kernel<HT, OT, NSB>), dim3(nblk, 1), dim3(256), lds,               \
                       osa_stream(stream), a);                                                    \
  } while (0)
  OSA_DISPATCH_OT(a.nd, OSA_CALL);
#undef OSA_CALL
  if (nblk > 1) {
    const int W = a.nd.P + OSA_NSTAT;
    hipLaunchKernelGGL(osa_slab_reduce_kernel, dim3((W + 255) / 256, 1), dim3(256), 0,
                       osa_stream(stream), a);
  }
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_fvp_finish(int n, const float* raw, const float* v, float damping, int ls_off, int ls_n,
                   float ls_coef, float* out, void* stream) {
  OSA_REQUIRE(n > 0 && raw && v && out);
  hipLaunchKernelGGL(osa_fvp_finish_kernel, dim3(1), dim3(1024), 0, osa_stream(stream), n, raw, v,
                     damping, ls_off, ls_n, ls_coef, out);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_cg_init(int n, const float* b, float* x, float* r, float* p, float* scal, void* stream) {
  OSA_REQUIRE(n > 0 && b && x && r && p && scal);
  hipLaunchKernelGGL(osa_cg_init_kernel, dim3(1), dim3(1024), 0, osa_stream(stream), n, b, x, r, p, scal);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_cg_step(int n, const float* z, float* x, float* r, float* p, float* scal, float residual_tol,
                float eps, void* stream) {
  OSA_REQUIRE(n > 0 && z && x && r && p && scal);
  hipLaunchKernelGGL(osa_cg_step_kernel, dim3(1), dim3(1024), 0, osa_stream(stream), n, z, x, r, p,
                     scal, residual_tol, eps);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_vec_lincomb(int n, float a, const float* x, float b, const float* y, float* out, void* stream) {
  OSA_REQUIRE(n > 0 && x && out);
  hipLaunchKernelGGL(osa_vec_lincomb_kernel, dim3((n + 1023) / 1024), dim3(1024), 0, osa_stream(stream),
                     n, a, x, b, y, out);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_vec_dot(int n, const float* x, const float* y, float* out, void* stream) {
  OSA_REQUIRE(n > 0 && x && y && out);
  hipLaunchKernelGGL(osa_vec_dot_kernel, dim3(1), dim3(1024), 0, osa_stream(stream), n, x, y, out);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_actor_eval(int obs_dim, int act_dim, int hidden, const float* actor_params, const float* obs,
                   int ld_obs, long M, const float* act, int ld_act, const float* logp,
                   const float* adv_r, const float* adv_c, const float* lagrange,
                   const float* old_mean, int ld_old, const float* old_log_std, double* ws,
                   float* out4, void* stream) {
  const int rc = osa_check_dims(obs_dim, act_dim, hidden);
  if (rc != OSA_OK) return rc;
  OSA_REQUIRE(actor_params && obs && act && logp && adv_r && adv_c && old_mean && old_log_std && ws && out4);
  OSA_REQUIRE(M > 0 && ld_obs >= obs_dim && ld_act >= act_dim && ld_old >= act_dim);
  const OsaNet nd = osa_make_net(obs_dim, act_dim, hidden);
  long nb = (M + 63) / 64;
  if (nb > 1024) nb = 1024;
#define OSA_CALL(HT, OT, NSB)                                                                          \
  hipLaunchKernelGGL((osa_actor_eval_kernel<HT, OT>), dim3((unsigned)nb), dim3(256), 0,           \
                     osa_stream(stream), nd, actor_params, obs, ld_obs, M, act, ld_act, logp,     \
                     adv_r, adv_c, lagrange, old_mean, ld_old, old_log_std, ws)
  OSA_DISPATCH_OT(nd, OSA_CALL);
#undef OSA_CALL
  hipLaunchKernelGGL(osa_eval_final_kernel, dim3(1), dim3(256), 0, osa_stream(stream), ws, (int)nb,
                     (double)M, (double)act_dim, out4);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_actor_kl(int obs_dim, int act_dim, int hidden, const float* actor_params, const float* obs,
                 int ld_obs, long M, const float* old_mean, int ld_old, const float* old_log_std,
                 int reduce_mode, float* mean_out, int ld_mean, double* ws, float* kl_out,
                 void* stream) {
  const int rc = osa_check_dims(obs_dim, act_dim, hidden);
  if (rc != OSA_OK) return rc;
  OSA_REQUIRE(actor_params && obs && M > 0 && ld_obs >= obs_dim);
  OSA_REQUIRE((old_mean == nullptr) || (old_log_std && ws && kl_out));
  const OsaNet nd = osa_make_net(obs_dim, act_dim, hidden);
  long nb = (M + 63) / 64;
  if (nb > 1024) nb = 1024;
#define OSA_CALL(HT, OT, NSB)                                                                          \
  hipLaunchKernelGGL((osa_actor_kl_kernel<HT, OT>), dim3((unsigned)nb), dim3(256), 0,             \
                     osa_stream(stream), nd, actor_params, obs, ld_obs, M, old_mean, ld_old,      \
                     old_log_std, mean_out, ld_mean, ws)
  OSA_DISPATCH_OT(nd, OSA_CALL);
#undef OSA_CALL
  if (old_mean) {
    const double denom = reduce_mode == 0 ? (double)M : (double)M * act_dim;
    hipLaunchKernelGGL(osa_kl_final_kernel, dim3(1), dim3(256), 0, osa_stream(stream), ws, (int)nb,
                       denom, kl_out);
  }
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

}  // extern "C"
