// Device building blocks for the actor-critic MLPs (Linear-tanh-Linear-tanh-Linear, reference
// omnisafe/utils/model.py:103-111) on gfx950 matrix cores.
//
// Everything is exact float32: v_mfma_f32_16x16x4_f32 is bit-for-bit an fmaf chain at the f32 vector
// rate (157 TFLOP/s peak), so results differ from the reference's CPU sgemm only by summation order.
//
// Orientation.  All products are computed TRANSPOSED: H^T[out x samples] = W[out x in] . X^T[in x
// samples].  With that choice the accumulator fragment of one layer IS the B-operand fragment of the
// next layer, register for register, so activations never leave VGPRs between layers:
//   MFMA 16x16x4:  A-frag lane l holds A[i = l&15][k = l>>4],  B-frag lane l holds B[k = l>>4][j = l&15],
//                  D-frag lane l holds D[i = 4*(l>>4) + r][j = l&15], r = 0..3.
// Within one 16-wide K block we permute K so that MFMA step s (0..3) of lane group g = l>>4 consumes
// k = 4g + s.  A-fragments are then 4 CONTIGUOUS floats of a weight row (one 16-byte load), and the
// D-fragment of the previous layer (rows 4g + r of lane group g) is exactly the 4 steps of the next
// layer's B-fragment.  ("S layout": lane = sample, registers = 4 consecutive features.)
//
// Weight gradients contract over samples instead of features and need the transposed ("F layout":
// lane = feature, registers = 4 consecutive samples) fragments; those go through LDS once.
#pragma once
#include "osa_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define OSA_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Padded parameter block of one network (all offsets in floats, multiples of 16 -> 64-byte aligned):
//   W1 [H][INP] | b1 [H] | W2 [H][H] | b2 [H] | W3 [OUTP][H] | b3 [OUTP] | log_std [OUTP]
// INP = obs_dim rounded up to 16, OUTP = max(act_dim, 1) rounded up to 16.  Padding entries are zero
// and stay zero (their gradients are identically zero).  The same block shape is used for the actor
// and both critics (critics use row 0 of W3 / b3[0]; their log_std slot is unused).
struct OsaNet {
  int obs_dim, act_dim, H, INP, OUTP, KB;
  int oW1, ob1, oW2, ob2, oW3, ob3, oLS, P;
  int act;  // hidden activation (OSA_ACT_*), decoded from the `hidden` word of the C ABI
};

// The `hidden` argument of every entry point: low 16 bits = width of the two hidden layers, bits 16-19 = their
// activation (reference model_cfgs.*.activation, utils/model.py:47-70): 0 tanh (default, every YAML), 1 relu,
// 2 sigmoid, 3 softplus, 4 identity.  The persistent pass kernels implement tanh only (osa_ppo_*_supported return 0
// otherwise: such networks run on the per-step kernels).
#define OSA_ACT_TANH 0
#define OSA_ACT_RELU 1
#define OSA_ACT_SIGMOID 2
#define OSA_ACT_SOFTPLUS 3
#define OSA_ACT_IDENTITY 4

__host__ __device__ inline OsaNet osa_make_net(int obs_dim, int act_dim, int hidden) {
  OsaNet n;
  const int H = hidden & 0xFFFF;
  n.act = (hidden >> 16) & 0xF;
  n.obs_dim = obs_dim;
  n.act_dim = act_dim;
  n.H = H;
  n.INP = (obs_dim + 15) / 16 * 16;
  n.OUTP = ((act_dim > 1 ? act_dim : 1) + 15) / 16 * 16;
  n.KB = n.INP / 16;
  n.oW1 = 0;
  n.ob1 = n.oW1 + H * n.INP;
  n.oW2 = n.ob1 + H;
  n.ob2 = n.oW2 + H * H;
  n.oW3 = n.ob2 + H;
  n.ob3 = n.oW3 + n.OUTP * H;
  n.oLS = n.ob3 + n.OUTP;
  n.P = n.oLS + n.OUTP;
  return n;
}

// tanh = 1 - 2 / (1 + e^{2x}) on the hardware exp2 / rcp units: 2 transcendental + 3 plain VALU ops
// per element, no branch, saturates correctly (e^{2x} -> inf gives 1, -> 0 gives -1).  Absolute error
// <= ~1.5e-7 (one ulp of 1.0) everywhere; ocml's tanhf costs ~5x as much and dominated the forward
// pass of the 64-row optimiser step (measured 2.3k of 8.2k cycles).  Relative error grows for
// |x| << 1 (the result is a difference of two numbers close to 1) but what feeds the next layer and
// the tanh' = 1 - h^2 factor is the ABSOLUTE value, so activations stay within float32 rounding of an
// O(1) quantity -- the same class of error as the summation-order differences vs the reference's sgemm.
__device__ __forceinline__ float osa_tanhf(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.88539008177792681472f);  // e^{2x} = 2^{2x log2 e}
  return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + e);
}
__device__ __forceinline__ f32x4 osa_tanh4(f32x4 v) {
#ifdef OSA_ABLATE_TANH
  return v * 0.5f;
#endif
  const f32x4 t = v * 2.88539008177792681472f;
  f32x4 e;
  e.x = __builtin_amdgcn_exp2f(t.x);
  e.y = __builtin_amdgcn_exp2f(t.y);
  e.z = __builtin_amdgcn_exp2f(t.z);
  e.w = __builtin_amdgcn_exp2f(t.w);
  e = e + 1.f;
  f32x4 r;
  r.x = __builtin_amdgcn_rcpf(e.x);
  r.y = __builtin_amdgcn_rcpf(e.y);
  r.z = __builtin_amdgcn_rcpf(e.z);
  r.w = __builtin_amdgcn_rcpf(e.w);
  return 1.f - 2.f * r;  // packed f32 math on the vector
}

// Hidden activation and its derivative EXPRESSED THROUGH THE OUTPUT h (what the backward pass has at hand):
//   tanh 1 - h^2 | relu [h > 0] | sigmoid h (1 - h) | softplus 1 - e^{-h} (= sigmoid(x)) | identity 1
// `act` is uniform over the launch: the switch is a scalar branch per tile, tanh stays the first (default) case.
__device__ __forceinline__ f32x4 osa_act4(f32x4 v, int act) {
  if (act == OSA_ACT_TANH) return osa_tanh4(v);
  f32x4 r;
  if (act == OSA_ACT_RELU) {
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = fmaxf(v[k], 0.f);
  } else if (act == OSA_ACT_SIGMOID) {
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = 1.f / (1.f + expf(-v[k]));
  } else if (act == OSA_ACT_SOFTPLUS) {  // torch.nn.Softplus(beta = 1, threshold = 20)
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = v[k] > 20.f ? v[k] : log1pf(expf(v[k]));
  } else {
    r = v;
  }
  return r;
}
__device__ __forceinline__ f32x4 osa_dact4(f32x4 h, int act) {
  if (act == OSA_ACT_TANH) return 1.f - h * h;
  f32x4 r;
  if (act == OSA_ACT_RELU) {
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = h[k] > 0.f ? 1.f : 0.f;
  } else if (act == OSA_ACT_SIGMOID) {
    r = h * (1.f - h);
  } else if (act == OSA_ACT_SOFTPLUS) {
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = 1.f - expf(-h[k]);
  } else {
    r = (f32x4){1.f, 1.f, 1.f, 1.f};
  }
  return r;
}

// One Adam step of one parameter (torch.optim.Adam single-tensor form):
//   m <- lerp(m, g, 1-b1); v <- b2 v + (1-b2) g^2; w <- w - step_size * m / (sqrt(v)/bc2_sqrt + eps)
// inv_bc2_sqrt = 1/sqrt(1 - b2^t), step_size = lr/(1 - b1^t) are formed once per step in float64.
// sqrt and the reciprocal use the hardware units (1 ulp): the relative error of the *update* is
// ~1e-7 of a step of size ~lr, i.e. far below float32 resolution of the parameter.
__device__ __forceinline__ float osa_adam_update(float g, float& m, float& v, float w, float beta1,
                                                 float beta2, float step_size, float inv_bc2_sqrt,
                                                 float eps) {
  m = fmaf(g - m, 1.f - beta1, m);
  v = fmaf(v, beta2, (1.f - beta2) * g * g);
  const float denom = fmaf(__builtin_amdgcn_sqrtf(v), inv_bc2_sqrt, eps);
  return fmaf(-step_size * m, __builtin_amdgcn_rcpf(denom), w);
}

// Four parameters at once (packed float32 math on the vector; sqrt / rcp per element).
__device__ __forceinline__ f32x4 osa_adam_update4(f32x4 g, f32x4& m, f32x4& v, f32x4 w, float beta1,
                                                  float beta2, float step_size, float inv_bc2_sqrt,
                                                  float eps) {
  m = m + (g - m) * (1.f - beta1);
  v = v * beta2 + (g * g) * (1.f - beta2);
  f32x4 sq;
  sq.x = __builtin_amdgcn_sqrtf(v.x);
  sq.y = __builtin_amdgcn_sqrtf(v.y);
  sq.z = __builtin_amdgcn_sqrtf(v.z);
  sq.w = __builtin_amdgcn_sqrtf(v.w);
  const f32x4 denom = sq * inv_bc2_sqrt + eps;
  f32x4 rc;
  rc.x = __builtin_amdgcn_rcpf(denom.x);
  rc.y = __builtin_amdgcn_rcpf(denom.y);
  rc.z = __builtin_amdgcn_rcpf(denom.z);
  rc.w = __builtin_amdgcn_rcpf(denom.w);
  return w - (m * step_size) * rc;
}

// X fragment (S layout): 4 consecutive input features [col0, col0+4) of this lane's sample row.
// `row` may be nullptr (masked sample) -> zeros.  vec_ok: rows are 16-byte aligned and ld % 4 == 0.
__device__ __forceinline__ f32x4 osa_load_x(const float* __restrict__ row, int col0, int obs_dim,
                                            int ld, bool vec_ok) {
  f32x4 x = {0.f, 0.f, 0.f, 0.f};
  if (row == nullptr || col0 >= obs_dim) return x;
  if (vec_ok && col0 + 4 <= ld) {
    x = *reinterpret_cast<const f32x4*>(row + col0);
    if (col0 + 1 >= obs_dim) x.y = 0.f;
    if (col0 + 2 >= obs_dim) x.z = 0.f;
    if (col0 + 3 >= obs_dim) x.w = 0.f;
  } else {
    x.x = row[col0];
    if (col0 + 1 < obs_dim) x.y = row[col0 + 1];
    if (col0 + 2 < obs_dim) x.z = row[col0 + 2];
    if (col0 + 3 < obs_dim) x.w = row[col0 + 3];
  }
  return x;
}

// The same fragment in two steps (round 6): osa_load_x zeroes the columns past obs_dim with selects ON THE LOADED DATA, and
// a select is a use -- behind a prefetch it makes the wave sit out the memory round trip at once (one trip per K block
// of the first layer instead of one for all; 1 600 cycles per chunk of the Fisher-vector product,
// profiles/r6_fvp_phase_clocks.txt).  osa_load_x_raw only REQUESTS (vec_ok: one 16-byte load from a clamped address,
// `safe` = any readable 16-byte aligned address for masked lanes), osa_mask_x turns the raw value into osa_load_x's.
__device__ __forceinline__ f32x4 osa_load_x_raw(const float* __restrict__ row, int col0, int obs_dim, int ld, bool vec_ok,
                                                const float* __restrict__ safe) {
  if (!vec_ok) return osa_load_x(row, col0, obs_dim, ld, false);
  // (ld % 4 == 0, col0 % 4 == 0 and col0 < obs_dim <= ld: the piece lies inside the row)
  const bool ok = row != nullptr && col0 < obs_dim;
  return *reinterpret_cast<const f32x4*>(ok ? row + col0 : safe);
}
__device__ __forceinline__ f32x4 osa_mask_x(f32x4 x, bool have_row, int col0, int obs_dim) {
  if (!(have_row && col0 < obs_dim)) x.x = 0.f;
  if (!(have_row && col0 + 1 < obs_dim)) x.y = 0.f;
  if (!(have_row && col0 + 2 < obs_dim)) x.z = 0.f;
  if (!(have_row && col0 + 3 < obs_dim)) x.w = 0.f;
  return x;
}

// Forward pass of one network for the 16 samples owned by this wave.
//   xrow : this lane's sample row (lane l -> sample l&15), nullptr if masked
//   h1,h2: hidden activations, S layout, HT tiles of 16 features
//   out  : output pre-activations, S layout, OT tiles (row 4g+r of tile o = output 16o+4g+r)
template <int HT, int OT>
__device__ __forceinline__ void osa_mlp_forward(const OsaNet& nd, const float* __restrict__ p,
                                                const float* __restrict__ xrow, int ld, bool vec_ok,
                                                f32x4 (&h1)[HT], f32x4 (&h2)[HT], f32x4 (&out)[OT]) {
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const int H = nd.H, INP = nd.INP;
  const float* __restrict__ W1 = p + nd.oW1;
  const float* __restrict__ W2 = p + nd.oW2;
  const float* __restrict__ W3 = p + nd.oW3;
#pragma unroll
  for (int t = 0; t < HT; ++t) h1[t] = *reinterpret_cast<const f32x4*>(p + nd.ob1 + 16 * t + 4 * g);
  // software-pipelined over the input K blocks: the x chunk and the four weight fragments of block kb+1 are
  // requested before the MFMAs of block kb issue (wide inputs -- Humanoid has 24 blocks -- otherwise pay one
  // L2 round trip per block)
  f32x4 xn = osa_load_x_raw(xrow, 4 * g, nd.obs_dim, ld, vec_ok, p);
  f32x4 wn[HT];
#pragma unroll
  for (int t = 0; t < HT; ++t) wn[t] = *reinterpret_cast<const f32x4*>(W1 + (long)(16 * t + i) * INP + 4 * g);
  for (int kb = 0; kb < nd.KB; ++kb) {
    const f32x4 x = osa_mask_x(xn, xrow != nullptr, 16 * kb + 4 * g, nd.obs_dim);
    f32x4 w[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) w[t] = wn[t];
    if (kb + 1 < nd.KB) {
      xn = osa_load_x_raw(xrow, 16 * (kb + 1) + 4 * g, nd.obs_dim, ld, vec_ok, p);
#pragma unroll
      for (int t = 0; t < HT; ++t)
        wn[t] = *reinterpret_cast<const f32x4*>(W1 + (long)(16 * t + i) * INP + 16 * (kb + 1) + 4 * g);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      h1[t] = OSA_MFMA(w[t].x, x.x, h1[t]);
      h1[t] = OSA_MFMA(w[t].y, x.y, h1[t]);
      h1[t] = OSA_MFMA(w[t].z, x.z, h1[t]);
      h1[t] = OSA_MFMA(w[t].w, x.w, h1[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int t = 0; t < HT; ++t) h1[t] = osa_act4(h1[t], nd.act);
#pragma unroll
  for (int t = 0; t < HT; ++t) h2[t] = *reinterpret_cast<const f32x4*>(p + nd.ob2 + 16 * t + 4 * g);
#pragma unroll
  for (int kb = 0; kb < HT; ++kb) {
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(W2 + (16 * t + i) * H + 16 * kb + 4 * g);
      h2[t] = OSA_MFMA(w.x, h1[kb].x, h2[t]);
      h2[t] = OSA_MFMA(w.y, h1[kb].y, h2[t]);
      h2[t] = OSA_MFMA(w.z, h1[kb].z, h2[t]);
      h2[t] = OSA_MFMA(w.w, h1[kb].w, h2[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < HT; ++t) h2[t] = osa_act4(h2[t], nd.act);
#pragma unroll
  for (int o = 0; o < OT; ++o) out[o] = *reinterpret_cast<const f32x4*>(p + nd.ob3 + 16 * o + 4 * g);
#pragma unroll
  for (int kb = 0; kb < HT; ++kb) {
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(W3 + (16 * o + i) * H + 16 * kb + 4 * g);
      out[o] = OSA_MFMA(w.x, h2[kb].x, out[o]);
      out[o] = OSA_MFMA(w.y, h2[kb].y, out[o]);
      out[o] = OSA_MFMA(w.z, h2[kb].z, out[o]);
      out[o] = OSA_MFMA(w.w, h2[kb].w, out[o]);
    }
  }
}

// Sum over the 4 lane groups g (same sample j in lanes j, j+16, j+32, j+48).
__device__ __forceinline__ float osa_sum_over_groups(float v) {
  // lanes l, l^16, l^32, l^48 (the four lane groups of a sample) -> their sum in all four.  gfx950's
  // v_permlane{16,32}_swap exchange row pairs inside the VALU: both results together are {own, partner}
  // (in either order), so their sum is v + v[l ^ 16] -- the same bits as the former ds_bpermute shuffles
  // without two dependent LDS-pipe round trips.
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// ---- Philox4x32-10 counter-based generator + Box-Muller (device-resident rollout noise) ------------
__device__ __forceinline__ void osa_philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0;
  c[1] = n1;
  c[2] = n2;
  c[3] = n3;
}
__device__ __forceinline__ void osa_philox(uint64_t seed, uint64_t ctr_hi, uint64_t ctr_lo,
                                           uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), (uint32_t)ctr_hi,
                   (uint32_t)(ctr_hi >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    osa_philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c[0];
  out[1] = c[1];
  out[2] = c[2];
  out[3] = c[3];
}
__device__ __forceinline__ float osa_u01(uint32_t x) {  // (0, 1]
  return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);
}
// two independent standard normals from two 32-bit words
__device__ __forceinline__ void osa_box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
  const float u1 = osa_u01(a), u2 = osa_u01(b);
  const float r = sqrtf(-2.0f * logf(u1));
  float s, c;
  sincosf(6.28318530717958647692f * u2, &s, &c);
  n0 = r * c;
  n1 = r * s;
}

// Defined in ppo_pass_kernel.hip: the gradient of one large minibatch on the persistent kernel's machinery
// (see there); slabs: [3][nblk][P + 16] partial gradients in the parameter layout.
int osa_pass_partial_grad(int obs_dim, int act_dim, int hidden, float* params, const float* obs, int ld_obs,
                          const float* act, int ld_act, const float* logp, const float* target_value_r,
                          const float* target_value_c, const float* adv_r, const float* adv_c,
                          const long* idx, int B, const float* lagrange, const osa_ppo_hparams* hp,
                          int loss_kind, int nets_mask, int nblk, float* slabs, void* stream);
// Balanced form (round 4): the chunk-tasks of all networks shared evenly by <= max_wg workgroups; per-network slab
// counts in nslab[3], slab stride in *stride.
int osa_pass_partial_grad_balanced(int obs_dim, int act_dim, int hidden, float* params, const float* obs, int ld_obs,
                                   const float* act, int ld_act, const float* logp, const float* target_value_r,
                                   const float* target_value_c, const float* adv_r, const float* adv_c,
                                   const long* idx, int B, const float* lagrange, const osa_ppo_hparams* hp,
                                   int loss_kind, int nets_mask, int max_wg, int max_stride, float* slabs,
                                   int* nslab, int* stride, void* stream);
