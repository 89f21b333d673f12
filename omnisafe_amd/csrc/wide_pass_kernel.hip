// Persistent PPO-Lag update pass for WIDE observations (97 .. 512 input features; BASELINE config 4:
// SafetyHumanoidVelocity, 376 / 17): ONE launch = one whole pass of PolicyGradient._update's inner loop
// (policy_gradient.py:366-382) -- `nmb` dependent minibatch optimiser steps of <= 64 rows -- for the three
// networks (blockIdx.x = network; one 256-thread workgroup = 4 waves x 16 samples).
//
// Why a second kernel: osa_ppo_pass_kernel keeps W1 in LDS and its Adam moments in registers, which stops at
// 96 inputs (6 K blocks).  With 376 inputs W1 alone is 96 KB and its moments 192 KB: they cannot live in one
// CU.  Round 1 therefore ran such networks on the per-step kernels: 3 launches per optimiser step, gradients
// written to and re-read from global memory, norm + Adam over 29 k parameters by one latency-bound block:
// 114 us per step.  Here the step stays inside one launch and one CU per network:
//   * W2, W3, biases, log_std: LDS master copy, Adam moments in registers (as in osa_ppo_pass_kernel);
//   * W1 and its moments stay in global memory but never leave the L2 (3 x 96 KB per network): the forward
//     pass streams W1 fragments with a one-block register prefetch, the W1 gradient is accumulated in
//     registers (one f32x4 MFMA accumulator per 16x16 tile: 24 tiles = 96 registers per lane), and every lane
//     applies norm contribution + clip + Adam to exactly the W1 elements its accumulators hold -- the gradient
//     never goes to memory;
//   * the [feature][sample] tiles the weight-gradient contraction needs (h1, h2, dz1, dz2, dout) go through
//     LDS once; the X^T operand of dW1 is restaged from the (L2-hot) observation rows in double-buffered slabs
//     of 4 K blocks.
// Arithmetic = osa_mb_grad_kernel + osa_finalize_net (same fragment algebra, same loss code), so the
// per-step path, this kernel and the reference agree to float32 summation order (tests/test_mlp_gpu.py,
// tests/test_config_shapes_gpu.py::config4).
//
// Roofline of one step and network (376/17): 6 W B = 6 x 29 232 x 64 = 11.2 MFLOP on one CU's matrix pipe
// = 1056 v_mfma_f32_16x16x4 per wave x 32 cycles = 33.8 k cycles = 14 us at 2.4 GHz; L2 traffic per step:
// W1 fragments 96 KB + (critics) W1 for the L2 term 96 KB + Adam 3 x 96 KB read and written.
#include "mlp_device.h"

#define WSLD 68  // leading dimension (floats) of every [row][64 + pad] LDS tile
#define WNSTAT 16

struct OsaWideHp {
  float clip, entropy_coef, critic_norm_coef, max_grad_norm;
  float lr_actor, lr_critic, beta1, beta2, adam_eps;
  int use_critic_norm, use_max_grad_norm, use_cost;
};

struct OsaWideArgs {
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
  OsaWideHp hp;
  int loss_kind;
  int nets_mask;
  float* stats;  // [nmb][WNSTAT]
};

// KB4 = ceil(KB / 4): the W1 gradient is held as 4 * KB4 accumulator tiles (tiles >= KB stay zero)
template <int KB4, int OT>
__global__ __launch_bounds__(256, 1) void osa_wide_pass_kernel(OsaWideArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int H = 64, HT = 4, OUTP = 16 * OT, KBM = 4 * KB4;
  const OsaNet& nd = a.nd;
  const int net = blockIdx.x;
  if (!((a.nets_mask >> net) & 1)) return;
  const int KB = nd.KB, INP = nd.INP, P = nd.P;
  // ---- LDS carve-up (offsets are multiples of 4 floats)
  float* sW2 = smem;                      // [H][WSLD]
  float* sW3 = sW2 + H * WSLD;            // [OUTP][WSLD]
  float* sB1 = sW3 + OUTP * WSLD;         // [H]
  float* sB2 = sB1 + H;                   // [H]
  float* sB3 = sB2 + H;                   // [OUTP]
  float* sLS = sB3 + OUTP;                // [OUTP]
  float* sH1 = sLS + OUTP;                // [H][WSLD]  element (feature f, sample c)
  float* sH2 = sH1 + H * WSLD;
  float* sZ1 = sH2 + H * WSLD;
  float* sZ2 = sZ1 + H * WSLD;
  float* sDO = sZ2 + H * WSLD;            // [OUTP][WSLD]
  float* sDL = sDO + OUTP * WSLD;         // [OUTP][WSLD]
  float* sXs = sDL + OUTP * WSLD;         // [2][64][WSLD]: X^T slab of 4 K blocks (64 input features), double-buffered
  float* red = sXs + 2 * 64 * WSLD;       // [16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int i = j, cc = j;
  const bool leader = tid == 192;
  float* __restrict__ gp = a.params + (long)net * P;
  float* __restrict__ gm = a.adam_m + (long)net * P;
  float* __restrict__ gv = a.adam_v + (long)net * P;
  float* __restrict__ W1 = gp + nd.oW1;
  float* __restrict__ M1 = gm + nd.oW1;
  float* __restrict__ V1 = gv + nd.oW1;
  const bool critic = net != 0;
  const bool vec_ok = (a.ld_obs % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.obs) & 15) == 0);

  // ---- small parameters -> LDS master copy
  for (int e = tid; e < H * H; e += 256) sW2[(e >> 6) * WSLD + (e & 63)] = gp[nd.oW2 + e];
  for (int e = tid; e < OUTP * H; e += 256) sW3[(e >> 6) * WSLD + (e & 63)] = gp[nd.oW3 + e];
  if (tid < H) {
    sB1[tid] = gp[nd.ob1 + tid];
    sB2[tid] = gp[nd.ob2 + tid];
  }
  if (tid < OUTP) {
    sB3[tid] = gp[nd.ob3 + tid];
    sLS[tid] = gp[nd.oLS + tid];
  }
  // ---- ownership (as osa_ppo_pass_kernel): W2[(16w+4g+r)][16ti+cc], W3[(16o+4g+r)][16w+cc], one
  // bias-like scalar per thread; W1[(16w+4g+r)][16kb+cc] lives in global memory
  f32x4 m2[HT], v2[HT], m3[OT], v3[OT];
  float mb_ = 0.f, vb_ = 0.f;
  int boff = -1;
  float* sbias = sB1;
  if (tid < H) { boff = nd.ob1 + tid; sbias = sB1 + tid; }
  else if (tid < 2 * H) { boff = nd.ob2 + tid - H; sbias = sB2 + tid - H; }
  else if (tid < 2 * H + OUTP) { boff = nd.ob3 + tid - 2 * H; sbias = sB3 + tid - 2 * H; }
  else if (tid < 2 * H + 2 * OUTP) { boff = nd.oLS + tid - 2 * H - OUTP; sbias = sLS + tid - 2 * H - OUTP; }
  if (critic && boff >= nd.oLS) boff = -1;  // critics have no log_std
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
  if (boff >= 0) {
    mb_ = gm[boff];
    vb_ = gv[boff];
  }
  const int step0 = a.adam_step[net];
  const float lr = critic ? a.hp.lr_critic : a.hp.lr_actor;
  const float beta1 = a.hp.beta1, beta2 = a.hp.beta2, aeps = a.hp.adam_eps;
  // Adam's bias corrections of every step of the launch, tabulated once (float64 like torch): columns
  // 10 + 2 net, 11 + 2 net of the step's statistics row
  for (int k = tid; k < a.nmb; k += 256) {
    const double t = (double)(step0 + k + 1);
    float* row = a.stats + (long)k * WNSTAT;
    row[10 + 2 * net] = (float)((double)lr / (1.0 - pow((double)beta1, t)));
    row[11 + 2 * net] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, t)));
  }
  const bool l2 = critic && a.hp.use_critic_norm;
  const float c2 = 2.f * a.hp.critic_norm_coef;
  float lam = 0.f;
  if (net == 0 && a.lagrange) lam = *a.lagrange;
  const float* __restrict__ tgt = (net == 1) ? a.tgt_r : a.tgt_c;
  // the W1 elements this lane owns (D layout of the weight-gradient tiles): rows 16 wave + 4 g + r, columns
  // 16 kb + cc
  const long w1row0 = (long)(16 * wave + 4 * g) * INP + cc;
  __syncthreads();  // LDS master copy + bias-correction table complete

#define WPUT_TILE(S, V, T)                                                           \
  _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) (S)[(16 * (T) + 4 * g + r_) * WSLD + c] = (V)[r_]

  for (int mb = 0; mb < a.nmb; ++mb) {
    const long mb_lo = (long)mb * a.B;
    const int Bcur = (int)(min(mb_lo + a.B, a.M) - mb_lo);
    const float invB = 1.f / (float)Bcur;
    const int c = 16 * wave + j;  // this lane's sample column
    const bool valid = c < Bcur;
    // invalid columns of a ragged minibatch gather a valid row (finite values) and meet dL/dout = 0
    const long pos = mb_lo + (valid ? c : 0);
    const long row = a.perm ? a.perm[pos] : pos;
    const float* __restrict__ xrow = a.obs + row * a.ld_obs;
    const float* bc_row = a.stats + (long)mb * WNSTAT + 10 + 2 * net;
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
    {  // layer 1: W1 fragments and x chunks streamed from global memory (L2), one K block ahead
      f32x4 xn = osa_load_x(xrow, 4 * g, nd.obs_dim, a.ld_obs, vec_ok);
      f32x4 wn[HT];
#pragma unroll
      for (int t = 0; t < HT; ++t) wn[t] = *reinterpret_cast<const f32x4*>(W1 + (long)(16 * t + i) * INP + 4 * g);
      for (int kb = 0; kb < KB; ++kb) {
        const f32x4 x = xn;
        f32x4 w[HT];
#pragma unroll
        for (int t = 0; t < HT; ++t) w[t] = wn[t];
        if (kb + 1 < KB) {
          xn = osa_load_x(xrow, 16 * (kb + 1) + 4 * g, nd.obs_dim, a.ld_obs, vec_ok);
#pragma unroll
          for (int t = 0; t < HT; ++t)
            wn[t] = *reinterpret_cast<const f32x4*>(W1 + (long)(16 * t + i) * INP + 16 * (kb + 1) + 4 * g);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < HT; ++t) h1[t] = OSA_MFMA(w[t][s], x[s], h1[t]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      h1[t] = osa_tanh4(h1[t]);
      WPUT_TILE(sH1, h1[t], t);
    }
#pragma unroll
    for (int t = 0; t < HT; ++t) h2[t] = *reinterpret_cast<const f32x4*>(sB2 + 16 * t + 4 * g);
#pragma unroll
    for (int kb = 0; kb < HT; ++kb) {
      f32x4 w[HT];
#pragma unroll
      for (int t = 0; t < HT; ++t) w[t] = *reinterpret_cast<const f32x4*>(sW2 + (16 * t + i) * WSLD + 16 * kb + 4 * g);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < HT; ++t) h2[t] = OSA_MFMA(w[t][s], h1[kb][s], h2[t]);
    }
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      h2[t] = osa_tanh4(h2[t]);
      WPUT_TILE(sH2, h2[t], t);
    }
#pragma unroll
    for (int o = 0; o < OT; ++o) out[o] = *reinterpret_cast<const f32x4*>(sB3 + 16 * o + 4 * g);
#pragma unroll
    for (int kb = 0; kb < HT; ++kb) {
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(sW3 + (16 * o + i) * WSLD + 16 * kb + 4 * g);
#pragma unroll
        for (int s = 0; s < 4; ++s) out[o] = OSA_MFMA(w[s], h2[kb][s], out[o]);
      }
    }
    // ================= loss, dL/d(out) (osa_mb_grad_kernel's code path without extensions) =================
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
      const float* __restrict__ arow = a.act + row * a.ld_act;
#pragma unroll
      for (int o = 0; o < OT; ++o) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = 16 * o + 4 * g + r;
          zv[o][r] = 0.f;
          ivar[o][r] = 0.f;
          if (d < nd.act_dim && valid) {
            const float sd = expf(sLS[d]);
            const float var = sd * sd;
            const float z = arow[d] - out[o][r];
            zv[o][r] = z;
            ivar[o][r] = 1.f / var;
            lp += -(z * z) / (2.f * var) - logf(sd) - 0.91893853320467274178f;
          }
        }
      }
      lp = osa_sum_over_groups(lp);
      const float ratio = valid ? expf(lp - a.logp[row]) : 0.f;
      if (valid) {
        const float adv = (a.adv_r[row] - lam * a.adv_c[row]) / (1.f + lam);  // ppo_lag.py:101-102
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
      const float diff = out[0][0] - tgt[row];
      if (g == 0) {
        loss_part = diff * diff;
        dO[0][0] = 2.f * diff * invB;
      }
    }
    // ================= backward through the hidden layers (S layout) =================
    f32x4 z2[HT], z1[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int o = 0; o < OT; ++o) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {  // A[i][k] = W3^T[16t+i][16o+4g+s]
          const float w = sW3[(16 * o + 4 * g + s) * WSLD + 16 * t + i];
          acc = OSA_MFMA(w, dO[o][s], acc);
        }
      }
      z2[t] = acc * (1.f - h2[t] * h2[t]);
      WPUT_TILE(sZ2, z2[t], t);
    }
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < HT; ++kb) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {  // A[i][k] = W2^T[16t+i][16kb+4g+s]
          const float w = sW2[(16 * kb + 4 * g + s) * WSLD + 16 * t + i];
          acc = OSA_MFMA(w, z2[kb][s], acc);
        }
      }
      z1[t] = acc * (1.f - h1[t] * h1[t]);
      WPUT_TILE(sZ1, z1[t], t);
    }
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      WPUT_TILE(sDO, dO[o], o);
      WPUT_TILE(sDL, dLS[o], o);
    }
    // first X^T slab of the dW1 contraction: requested before the barrier, staged after it
    f32x4 xs[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) xs[q] = osa_load_x(xrow, 16 * q + 4 * g, nd.obs_dim, a.ld_obs, vec_ok);
    __syncthreads();  // (A) tiles complete
    // ================= weight gradients (registers) =================
    f32x4 g2[HT], g3[OT], g1[KBM];
    f32x4 a1[4];
    {
      f32x4 a2[4];
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        a2[sb] = *reinterpret_cast<const f32x4*>(sZ2 + (16 * wave + i) * WSLD + 16 * sb + 4 * g);
        a1[sb] = *reinterpret_cast<const f32x4*>(sZ1 + (16 * wave + i) * WSLD + 16 * sb + 4 * g);
      }
#pragma unroll
      for (int ti = 0; ti < HT; ++ti) g2[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        f32x4 b[HT];
#pragma unroll
        for (int ti = 0; ti < HT; ++ti)
          b[ti] = *reinterpret_cast<const f32x4*>(sH1 + (16 * ti + i) * WSLD + 16 * sb + 4 * g);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int ti = 0; ti < HT; ++ti) g2[ti] = OSA_MFMA(a2[sb][s], b[ti][s], g2[ti]);
      }
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        g3[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(sDO + (16 * o + i) * WSLD + 16 * sb + 4 * g);
          const f32x4 b = *reinterpret_cast<const f32x4*>(sH2 + (16 * wave + i) * WSLD + 16 * sb + 4 * g);
#pragma unroll
          for (int s = 0; s < 4; ++s) g3[o] = OSA_MFMA(av[s], b[s], g3[o]);
        }
      }
    }
    // bias-like gradient owned by this thread: row sum over the 64 samples
    float gb = 0.f;
    {
      const float* srow = (tid < H) ? sZ1 + tid * WSLD
                          : (tid < 2 * H) ? sZ2 + (tid - H) * WSLD
                          : (tid < 2 * H + OUTP) ? sDO + (tid - 2 * H) * WSLD
                          : (tid < 2 * H + 2 * OUTP) ? sDL + (tid - 2 * H - OUTP) * WSLD
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
    // ---- dW1[f][k] = sum_s dz1[f][s] x[s][k]: X^T restaged in slabs of 4 K blocks (64 input features),
    // double-buffered; the next slab's rows are requested before the current slab's MFMAs issue
#pragma unroll
    for (int s4 = 0; s4 < KB4; ++s4) {
      float* slab = sXs + (s4 & 1) * 64 * WSLD;
      if (4 * s4 < KB) {  // block-uniform
#pragma unroll
        for (int q = 0; q < 4; ++q) WPUT_TILE(slab, xs[q], q);
      }
      __syncthreads();
      if (4 * (s4 + 1) < KB) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          xs[q] = osa_load_x(xrow, 16 * (4 * (s4 + 1) + q) + 4 * g, nd.obs_dim, a.ld_obs, vec_ok);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kb = 4 * s4 + q;
        g1[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (kb < KB) {
#pragma unroll
          for (int sb = 0; sb < 4; ++sb) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(slab + (16 * q + i) * WSLD + 16 * sb + 4 * g);
#pragma unroll
            for (int s = 0; s < 4; ++s) g1[kb] = OSA_MFMA(a1[sb][s], b[s], g1[kb]);
          }
        }
      }
    }
    // ================= + 2 coef w (critics), squared norms =================
    f32x4 w2r[HT], w3r[OT];
#pragma unroll
    for (int ti = 0; ti < HT; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) w2r[ti][r] = sW2[(16 * wave + 4 * g + r) * WSLD + 16 * ti + cc];
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r) w3r[o][r] = sW3[(16 * o + 4 * g + r) * WSLD + 16 * wave + cc];
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
    if (critic) {  // block-uniform: the L2 term and sum p^2 need W1's values (L2 hits)
#pragma unroll
      for (int kb = 0; kb < KBM; ++kb) {
        if (kb < KB) {
          f32x4 w;
#pragma unroll
          for (int r = 0; r < 4; ++r) w[r] = W1[w1row0 + (long)r * INP + 16 * kb];
          if (l2) g1[kb] = g1[kb] + w * c2;
          acc_p = acc_p + w * w;
          acc_g = acc_g + g1[kb] * g1[kb];
        }
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < KBM; ++kb) acc_g = acc_g + g1[kb] * g1[kb];
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
    __syncthreads();  // (B)
    const float t_gsq = red[0] + red[4] + red[8] + red[12];
    const float t_psq = red[1] + red[5] + red[9] + red[13];
    const float t_loss = red[2] + red[6] + red[10] + red[14];
    const float t_ratio = red[3] + red[7] + red[11] + red[15];
    const float total_norm = sqrtf(t_gsq);
    float gscale = 1.f;
    if (a.hp.use_max_grad_norm) {
      gscale = a.hp.max_grad_norm / (total_norm + 1e-6f);
      gscale = gscale > 1.f ? 1.f : gscale;
    }
    // ================= Adam =================
    // W1: this lane's 4 x KB elements, read-modify-write in global memory (L2-resident)
#pragma unroll
    for (int kb = 0; kb < KBM; ++kb) {
      if (kb < KB) {
        f32x4 w, m, v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long off = w1row0 + (long)r * INP + 16 * kb;
          w[r] = W1[off];
          m[r] = M1[off];
          v[r] = V1[off];
        }
        w = osa_adam_update4(g1[kb] * gscale, m, v, w, beta1, beta2, step_size, inv_bc2_sqrt, aeps);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long off = w1row0 + (long)r * INP + 16 * kb;
          W1[off] = w[r];
          M1[off] = m[r];
          V1[off] = v[r];
        }
      }
    }
#pragma unroll
    for (int ti = 0; ti < HT; ++ti) {
      const f32x4 w = osa_adam_update4(g2[ti] * gscale, m2[ti], v2[ti], w2r[ti], beta1, beta2, step_size,
                                       inv_bc2_sqrt, aeps);
#pragma unroll
      for (int r = 0; r < 4; ++r) sW2[(16 * wave + 4 * g + r) * WSLD + 16 * ti + cc] = w[r];
    }
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      const f32x4 w = osa_adam_update4(g3[o] * gscale, m3[o], v3[o], w3r[o], beta1, beta2, step_size,
                                       inv_bc2_sqrt, aeps);
#pragma unroll
      for (int r = 0; r < 4; ++r) sW3[(16 * o + 4 * g + r) * WSLD + 16 * wave + cc] = w[r];
    }
    if (boff >= 0) {
      float mv_ = mb_, vv_ = vb_;
      *sbias = osa_adam_update(gb * gscale, mv_, vv_, wb, beta1, beta2, step_size, inv_bc2_sqrt, aeps);
      mb_ = mv_;
      vb_ = vv_;
    }
    if (leader) {
      float* st = a.stats + (long)mb * WNSTAT;
      if (net == 0) {
        st[2] = t_loss * invB - a.hp.entropy_coef * ent_pre;
        st[3] = t_ratio * invB;
        st[4] = ent_pre;
        st[7] = total_norm;
      } else {
        st[net - 1] = t_loss * invB;
        st[4 + net] = t_psq;
        st[7 + net] = total_norm;
      }
    }
    // (C) frees tiles and `red`; it also orders this step's W1 stores before the next step's W1 fragment
    // loads by the other waves: workgroup scope suffices, the four waves share one CU and its vector L1
    __syncthreads();
  }
#undef WPUT_TILE
  // ---- write back the LDS master copy and the register-resident Adam state
  for (int e = tid; e < H * H; e += 256) gp[nd.oW2 + e] = sW2[(e >> 6) * WSLD + (e & 63)];
  for (int e = tid; e < OUTP * H; e += 256) gp[nd.oW3 + e] = sW3[(e >> 6) * WSLD + (e & 63)];
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

static size_t osa_wide_lds_bytes(int OT) {
  const int H = 64, OUTP = 16 * OT;
  const size_t fl = (size_t)H * WSLD + (size_t)OUTP * WSLD + 2 * H + 2 * OUTP + 4 * (size_t)H * WSLD +
                    2 * (size_t)OUTP * WSLD + 2 * 64 * (size_t)WSLD + 64;
  return fl * sizeof(float);
}

template <int KB4, int OT>
static int osa_launch_wide(const OsaWideArgs& a, hipStream_t stream) {
  static bool attr_set = false;
  const size_t lds = osa_wide_lds_bytes(OT);
  if (lds > 160 * 1024) return OSA_EUNSUPPORTED;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&osa_wide_pass_kernel<KB4, OT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return OSA_EHIP;
    attr_set = true;
  }
  hipLaunchKernelGGL((osa_wide_pass_kernel<KB4, OT>), dim3(3), dim3(256), lds, stream, a);
  return hipGetLastError() == hipSuccess ? OSA_OK : OSA_EHIP;
}

extern "C" {

int osa_ppo_wide_pass_supported(int obs_dim, int act_dim, int hidden) {
  if (hidden != 64 || obs_dim < 97 || obs_dim > 512 || act_dim < 1 || act_dim > 32) return 0;
  return 1;
}

int osa_ppo_wide_pass(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                      int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                      const float* logp, const float* target_value_r, const float* target_value_c,
                      const float* adv_r, const float* adv_c, const long* perm, long M, int B,
                      const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                      float* step_stats, void* stream) {
  if (!osa_ppo_wide_pass_supported(obs_dim, act_dim, hidden)) return OSA_EUNSUPPORTED;
  if (B > 64 || loss_kind < 0 || loss_kind > 1) return OSA_EUNSUPPORTED;  // larger batches: per-step kernels
  OSA_REQUIRE(params && adam_m && adam_v && adam_step && obs && act && logp && hp && step_stats);
  OSA_REQUIRE(target_value_r && target_value_c && adv_r && adv_c && M > 0 && B > 0);
  OSA_REQUIRE(ld_obs >= obs_dim && ld_act >= act_dim);
  OsaWideArgs a = {};
  a.nd = osa_make_net(obs_dim, act_dim, hidden);
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
  const int KB4 = (a.nd.KB + 3) / 4, OT = a.nd.OUTP / 16;
#define OSA_WIDE_CASE(K, O) \
  if (KB4 == K && OT == O) return osa_launch_wide<K, O>(a, st)
  OSA_WIDE_CASE(2, 1); OSA_WIDE_CASE(3, 1); OSA_WIDE_CASE(4, 1); OSA_WIDE_CASE(5, 1); OSA_WIDE_CASE(6, 1);
  OSA_WIDE_CASE(7, 1); OSA_WIDE_CASE(8, 1);
  OSA_WIDE_CASE(2, 2); OSA_WIDE_CASE(3, 2); OSA_WIDE_CASE(4, 2); OSA_WIDE_CASE(5, 2); OSA_WIDE_CASE(6, 2);
  OSA_WIDE_CASE(7, 2); OSA_WIDE_CASE(8, 2);
#undef OSA_WIDE_CASE
  return OSA_EUNSUPPORTED;
}

}  // extern "C"
