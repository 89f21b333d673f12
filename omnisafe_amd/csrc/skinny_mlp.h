// Skinny kernels of the layer-wise path for general networks (general_mlp.hip): minibatches of at most 64 rows -- the
// reference's YAML batch_size (configs/on-policy/PPOLag.yaml: batch_size 64) -- on networks too wide for one compute
// unit (utils/model.py:73-111 builds any hidden_sizes; docs/source/start/efficiency.rst:15-23 times 1024 x 1024).
// Own header so that tools/skinny_probe.hip can build the kernels alone.
#pragma once
#include "mlp_device.h"

#define GM_MAXL OSA_GMLP_MAX_LAYERS

namespace {

// ---- activations (scalar forms of mlp_device.h's)
__device__ __forceinline__ float gm_act(float v, int act) {
  if (act == OSA_ACT_TANH) return osa_tanhf(v);
  if (act == OSA_ACT_RELU) return fmaxf(v, 0.f);
  if (act == OSA_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  if (act == OSA_ACT_SOFTPLUS) return v > 20.f ? v : log1pf(expf(v));
  return v;
}
__device__ __forceinline__ float gm_dact(float h, int act) {  // derivative through the OUTPUT h
  if (act == OSA_ACT_TANH) return 1.f - h * h;
  if (act == OSA_ACT_RELU) return h > 0.f ? 1.f : 0.f;
  if (act == OSA_ACT_SIGMOID) return h * (1.f - h);
  if (act == OSA_ACT_SOFTPLUS) return 1.f - expf(-h);
  return 1.f;
}

__device__ __forceinline__ f32x4 gm_load4(const float* __restrict__ row, int c0, int limit, bool ok) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (!ok || c0 >= limit) return v;
  if (c0 + 3 < limit) return *reinterpret_cast<const f32x4*>(row + c0);
  v.x = row[c0];
  if (c0 + 1 < limit) v.y = row[c0 + 1];
  if (c0 + 2 < limit) v.z = row[c0 + 2];
  return v;
}

__device__ __forceinline__ float gm_block_sum(float v, float* red) {  // deterministic; result to all; 256 threads
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = osa_wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  const float s = (red[0] + red[1]) + (red[2] + red[3]);
  return s;
}

// ------------------------------------------------------------------------------------------------
// loss and dL/d(output) of the layer-wise path
// ------------------------------------------------------------------------------------------------
struct GLossArgs {
  long R;
  int act_dim, lda, ldo[3];          // row strides of the action rows and of the three output layers
  const float* out[3];               // output-layer rows of the three networks [R][ldo]
  float* dz[3];                      // dL/d(output) [R][ldz]
  int ldz[3];
  const float* actg;
  const float* scal;                 // [0] logp [1] adv_r [2] adv_c [3] target_value_r [4] target_value_c
  const float* log_std;              // actor's log_std [act_dim]
  const float* lagrange;
  float clip;
  int loss_kind, nets_mask;
  float* dls;                        // [nblk][lda] per-block sums of dL/d(log_std)
  float* lpart;                      // [3][nblk][4] per-block {loss, ratio} sums
  int nblk;
  // Fisher-vector product (loss_kind 2): dL/d(out) = tangent of the mean / sigma^2 * fvp_scale
  const float* tmean;
  int ldt;
  float fvp_scale;
  // extended actor surrogates (osa_surrogate_ext: FOCOPS, CUP's second stage, P3O); ext_on = 0: none
  int ext_on;
  const long* idx;           // minibatch rows (old_mean is indexed like obs), or nullptr
  const float* old_mean;
  int ld_old_mean;
  const float* old_log_std;
  float ext_kl_coef, ext_mask_eta, ext_ratio_scale, ext_cost_kappa, ext_cost_excess;
  float* stats;              // stats[10] receives P3O's penalty value
  // direct != 0 (skinny path: no gather launch): the per-sample scalars and the action rows are read from the caller's
  // arrays through the minibatch's row indices -- sp[k][row], act[row * ld_act + d], row = idx ? idx[b] : b
  int direct, ld_act;
  const float* sp[5];
  const float* act;
};

// grid (nblk, 3): one thread per row.  Actor (policy_gradient.py:514-524 with PPOLag's surrogate, ppo.py:66-87 /
// policy_gradient.py:574-578) and critics (policy_gradient.py:428-433: mean squared error; the L2 term joins in
// gm_reduce_kernel).
// (a device function of 256 threads: gm_loss_kernel = one block per 256 rows and network; gs_top_kernel calls it for the
// <= 64 rows of a skinny step in EVERY workgroup of the top layer's backward launch -- dzp / ldzp then is an LDS image and
// only the `writer` workgroup leaves the block partials and statistics)
// The per-sample operands come through an accessor: GLossMem reads everything through GLossArgs (tiled path, round 5's
// top launch); GLossLds is round 6's top launch, whose operands the caller has staged in LDS and registers -- explicit
// LDS pointers, because a float* that MAY point to LDS is a flat access (~10 x an LDS read, and the loop over the
// action dimensions re-reads log_std from memory every iteration).
// WAVE: called by wave 0 alone for <= 64 rows, the sums are wave sums (no __syncthreads).
// sums != nullptr: thread 0 leaves {loss sum, ratio sum, dL/d(log_std)[0 .. act_dim)} of the block there.
typedef __attribute__((address_space(3))) float lds_f;
#ifdef GS_CLOCKS  // (tools/skinny_probe.hip: phase stamps inside the loss of workgroup 1)
__device__ long long gs_lclk[16];
#define GS_LSTAMP(i) do { if (WAVE && threadIdx.x == 0 && blockIdx.x == 1) gs_lclk[i] = __builtin_readcyclecounter(); } while (0)
#else
#define GS_LSTAMP(i) do { } while (0)
#endif
struct GLossMem {
  const GLossArgs& a;
  long b, row;
  const float* outp;
  float* dzp;
  __device__ __forceinline__ float sc(int k) const { return a.direct ? a.sp[k][row] : a.scal[(long)k * a.R + b]; }
  __device__ __forceinline__ float act(int d) const { return a.direct ? a.act[row * a.ld_act + d] : a.actg[b * a.lda + d]; }
  __device__ __forceinline__ float out(int d) const { return outp[d]; }
  __device__ __forceinline__ float ls(int d) const { return a.log_std[d]; }
  __device__ __forceinline__ float sd(int d) const { return expf(a.log_std[d]); }
  __device__ __forceinline__ float logsd(int d) const { return logf(sd(d)); }
  __device__ __forceinline__ float iv(int d) const { const float s_ = sd(d); return 1.f / (s_ * s_); }
  __device__ __forceinline__ float lam() const { return a.lagrange ? *a.lagrange : 0.f; }
  __device__ __forceinline__ void dz(int d, float v) const { dzp[d] = v; }
};
struct GLossLds {
  float sc_[5], lam_;
  const lds_f *act_, *out_, *ls_;  // this row's actions and outputs; log_std [0 .. 32), then the tables (the same
                                   // float32 expressions, evaluated once per workgroup instead of per row and loop):
                                   // exp(log_std) [32 .. 64), log(exp(log_std)) [64 .. 96), 1 / sd^2 [96 .. 128)
  lds_f* dz_;                      // this row of the dL/d(output) image
  __device__ __forceinline__ float sc(int k) const {
    return k == 0 ? sc_[0] : (k == 1 ? sc_[1] : (k == 2 ? sc_[2] : (k == 3 ? sc_[3] : sc_[4])));
  }
  __device__ __forceinline__ float act(int d) const { return act_[d]; }
  __device__ __forceinline__ float out(int d) const { return out_[d]; }
  __device__ __forceinline__ float ls(int d) const { return ls_[d]; }
  __device__ __forceinline__ float sd(int d) const { return ls_[32 + d]; }
  __device__ __forceinline__ float logsd(int d) const { return ls_[64 + d]; }
  __device__ __forceinline__ float iv(int d) const { return ls_[96 + d]; }
  __device__ __forceinline__ float lam() const { return lam_; }
  __device__ __forceinline__ void dz(int d, float v) const { dz_[d] = v; }
};
template <bool WAVE, typename ACC>
__device__ __forceinline__ void gm_loss_body(const GLossArgs& a, const int net, const long b, const int blk, float* red,
                                             const int ldzp, const bool writer, const ACC& acc, float* sums,
                                             const long row) {
  const bool valid = b < a.R;
  auto SC = [&](int k) -> float { return acc.sc(k); };
  auto ACT = [&](int d) -> float { return acc.act(d); };
  auto BSUM = [&](float v) -> float { return WAVE ? osa_wave_sum_dpp(v) : gm_block_sum(v, red); };
  const float invB = 1.f / (float)a.R;
  float loss = 0.f, ratio_s = 0.f;
    GS_LSTAMP(0);
  if (net != 0) {
    if (valid) {
      const float diff = acc.out(0) - SC(net == 1 ? 3 : 4);
      loss = diff * diff;
      acc.dz(0, 2.f * diff * invB);
      for (int d = 1; d < ldzp; ++d) acc.dz(d, 0.f);  // (row padding: the skinny kernels' 16-byte loads)
    }
  } else if (a.loss_kind == 2) {
    if (valid)
      for (int d = 0; d < a.act_dim; ++d) {
        const float sd = acc.sd(d);
        acc.dz(d, a.tmean[b * a.ldt + d] / (sd * sd) * a.fvp_scale);
      }
    if (valid)
      for (int d = a.act_dim; d < ldzp; ++d) acc.dz(d, 0.f);
  } else {
    const float lam = acc.lam();
    float lp = 0.f;
    if (valid)
      for (int d = 0; d < a.act_dim; ++d) {
        const float sd = acc.sd(d);
        const float z = ACT(d) - acc.out(d);
        lp += -(z * z) / (2.f * (sd * sd)) - acc.logsd(d) - 0.91893853320467274178f;
      }
    const float ratio = valid ? expf(lp - SC(0)) : 0.f;
    GS_LSTAMP(1);
    // ---- extended surrogates: per-sample KL(pi_theta || pi_old) (torch.distributions.kl._kl_normal_normal), FOCOPS'
    // trust mask with the reference's broadcast semantics (focops.py:84-88: the surrogate term sees the minibatch MEAN
    // of the mask), P3O's kappa * relu(mean(ratio * A_c) + excess) -- the arithmetic of osa_mb_grad_kernel's EXT form.
    // The mask mean and the penalty are minibatch-level: ONE block (the entry point refuses more than 256 rows).
    float kl = 0.f, mask = 1.f, mask_mean = 1.f, cost_w = 0.f;
    const long orow = row;
    if (a.ext_on) {
      if (valid)
        for (int d = 0; d < a.act_dim; ++d) {
          const float ls = acc.ls(d), ls0 = a.old_log_std[d], dl = ls - ls0;
          const float q = expf(dl), isd0 = expf(-ls0);
          const float u = (acc.out(d) - a.old_mean[orow * a.ld_old_mean + d]) * isd0;
          kl += 0.5f * (q * q + u * u - 1.f - 2.f * dl);
        }
      if (a.ext_mask_eta >= 0.f || a.ext_cost_kappa > 0.f) {  // block-uniform
        mask = (a.ext_mask_eta < 0.f || (valid && kl <= a.ext_mask_eta)) ? 1.f : 0.f;
        const float tm = BSUM(valid ? mask : 0.f);
        const float tc = BSUM(valid ? ratio * SC(2) : 0.f);
        if (a.ext_mask_eta >= 0.f) mask_mean = tm * invB;
        if (a.ext_cost_kappa > 0.f) {
          const float pen = tc * invB + a.ext_cost_excess;
          if (pen > 0.f) cost_w = a.ext_cost_kappa;
          if (threadIdx.x == 0 && a.stats && writer) a.stats[10] = a.ext_cost_kappa * fmaxf(pen, 0.f);
        }
      }
    }
    float dlogp = 0.f, dklw = 0.f;
    if (valid) {
      const float adv = (SC(1) - lam * SC(2)) / (1.f + lam);
      float dratio;
      if (a.loss_kind == 0) {
        const float lo = 1.f - a.clip, hi = 1.f + a.clip;
        const float rc = fminf(fmaxf(ratio, lo), hi);
        const float s1 = ratio * adv, s2 = rc * adv;
        const bool inrange = ratio >= lo && ratio <= hi;
        loss = -fminf(s1, s2);
        dratio = (s1 < s2 || inrange) ? -adv : 0.f;
      } else {
        loss = -(ratio * adv);
        dratio = -adv;
      }
      if (a.ext_on) {
        const float rs = a.ext_ratio_scale * mask_mean;
        loss = loss * rs + a.ext_kl_coef * kl * mask;
        dratio = dratio * rs + cost_w * SC(2);
        dklw = a.ext_kl_coef * mask * invB;
      }
      ratio_s = ratio;
      dlogp = dratio * ratio * invB;
    }
    GS_LSTAMP(2);
    // d logp / d mu = z / var;  d logp / d log_std = z^2 / var - 1; block sums of the latter, dimension by dimension
    for (int d = 0; d < a.act_dim; ++d) {
      float dl = 0.f;
      if (valid) {
        const float iv = acc.iv(d);
        const float mu = acc.out(d);
        const float z = ACT(d) - mu;
        float dmu = dlogp * z * iv;
        dl = dlogp * (z * z * iv - 1.f);
        if (a.ext_on) {  // d KL / d mu = (mu - mu0) / var0;  d KL / d log_std = var / var0 - 1
          const float ls0 = a.old_log_std[d], q = expf(acc.ls(d) - ls0), isd0 = expf(-ls0);
          const float u = (mu - a.old_mean[orow * a.ld_old_mean + d]) * isd0;
          dmu += dklw * (u * isd0);
          dl += dklw * (q * q - 1.f);
        }
        acc.dz(d, dmu);
      }
      dl = BSUM(dl);
      if (threadIdx.x == 0 && writer) a.dls[(long)blk * a.lda + d] = dl;
      if (threadIdx.x == 0 && sums) sums[2 + d] = dl;
    }
    GS_LSTAMP(3);
    if (valid)
      for (int d = a.act_dim; d < ldzp; ++d) acc.dz(d, 0.f);  // (row padding)
  }
    GS_LSTAMP(4);
  loss = BSUM(loss);
  ratio_s = BSUM(ratio_s);
    GS_LSTAMP(5);
  if (threadIdx.x == 0 && writer) {
    float* lp_ = a.lpart + ((long)net * a.nblk + blk) * 4;
    lp_[0] = loss;
    lp_[1] = ratio_s;
  }
  if (threadIdx.x == 0 && sums) {
    sums[0] = loss;
    sums[1] = ratio_s;
  }
  GS_LSTAMP(6);
}


// ------------------------------------------------------------------------------------------------
// small minibatches (rows <= 64: the reference's YAML batch_size): the skinny path
// ------------------------------------------------------------------------------------------------
// A 64-row optimiser step at hidden 1024 is bandwidth work -- 13 MB of weights per network pass, 2 x 64 flops per weight
// byte read -- that the tiled GEMM above runs as 17 launches of 10-30 us on a few dozen workgroups each.  Here:
//   gs_fwd_kernel    Y[r][n] = act(sum_k X[r][k] W[n][k] + b[n]): one workgroup per 16 output columns and ALL rows,
//                    8 waves split the contraction; the weights are streamed ONCE, 16 bytes per lane with the whole
//                    range of a wave in flight before its first MFMA, the activations come from L2;
//                    layer 0 reads the minibatch's rows in place through the row indices (no gather launch);
//   gs_top_kernel    loss + backward through the top layer in ONE launch: every workgroup computes dL/d(output) of all
//                    <= 64 rows itself in LDS (gm_loss_body), workgroup 0 of a network writes what later launches need;
//   gs_bwd_kernel    dZ'[r][k] = (sum_n dZ[r][n] W[n][k]) act'(H[r][k]): one workgroup per 16 columns k, the waves
//                    split the contraction over n;
//   gs_wgrad_kernel  dW = dZ^T H of ALL layers and networks in one launch, one 64 x 64 tile per workgroup (operands via
//                    LDS, the contraction is over <= 64 rows): phase 0 adds the critics' L2 term and leaves squared-norm
//                    partials (and the gradient itself only where the caller asks for it), phase 1 REcomputes the tile
//                    -- 64 MFMAs per wave -- and applies clip + Adam straight from the accumulators: the 13 MB gradient
//                    is never written or read, a step streams weights twice and the Adam state once.
// 2 L + 1 launches per step for L linear layers.  Same arithmetic per element as the tiled path up to the summation order
// of the contraction (float32 MFMA chains).  What a launch of gs_fwd / gs_bwd on a 1024 x 1024 layer costs is the L2 -> L1
// traffic of the 64 input rows, which every one of its 192 workgroups reads (tools/skinny_probe.hip; a split-K form that
// avoids it loses more to the exchange of partial tiles between workgroups: profiles/HISTORY.md, round-5 table).
struct GSProb {
  const float* X;     // fwd: input rows [R][ldx];  bwd: dZ rows [R][ldx]
  const float* W;     // [N][ldw]
  const float* bias;  // fwd: [N]
  const float* aux;   // bwd: stored outputs of the layer below [R][ldaux]
  float* Y;
  int ldx, ldw, ldy, ldaux;
  int N, K;           // W is N x K
  int act;            // fwd: activation, or -1;  bwd: activation whose derivative multiplies, or -1
  int net;            // the network this problem belongs to (gs_top_kernel)
  // ---- round 6: the clip norm without a weight-gradient launch of its own (GSArgs.fused != 0; see gs_gram_norm)
  float* Ylin;        // fwd: the layer's outputs before bias and activation [R][ldy] (critics with the L2 term), or null
  float* slot;        // fwd: norm slots of the tile [tiles][4] = {0, sum b^2, sum w^2, 0} of its 16 weight rows, or null
  const float* W2;    // fwd: the network's TOP layer fused into this launch: its weights [N2][ldw2] ...
  float* oslab;       //      ... and the partial outputs of the tile's 16 columns [tiles][64][ldo2]; top: the same slabs
  int ldw2, N2, ldo2, nbo;   // (nbo: tiles that wrote a slab)
  int wrows;          // fwd<true>: W's rows are gathered like X's (Gram matrix of the observation rows)
  const float* G;     // bwd / top: Gram matrix [64][64] of the input rows of the layer whose dZ this launch produces
  const float* Zlin;  // bwd / top: that layer's Ylin, or null
  const float* bvec;  // bwd / top: that layer's bias
  float* nslot;       // bwd / top: norm slots [tiles][4] = {partial squared gradient norm, 0, 0, 0}
  const float* btop;  // top: the top layer's bias
  float* tslot;       // top: norm slots of the top layer [tiles + 1][4] = {squared gradient norm, sum w^2 + b^2, 0, 0}
  float c2;           // 2 critic_norm_coef for a critic with the L2 term, else 0
};
#define GS_MAXPROB 6
struct GSArgs {
  GSProb p[GS_MAXPROB];
  int nprob, R;
  // != 0 (forward of layer 0, round 5: no gather launch): X is the CALLER's observation array -- any row stride, any
  // alignment -- and row r of the minibatch is X[(xidx ? xidx[r] : r) * ldx ..]: four-byte loads, K = obs_dim is small
  int xrows;
  const long* xidx;
  int fused;          // round 6: norm partials, fused top layer (mode 0 only)
  // round 6: ONE grid dimension -- workgroup b serves tile b - tile0[y] of problem y (tile0[y] <= b < tile0[y + 1]).  A
  // (tiles, problems) grid whose problems differ in size is padded with workgroups that exit at once, and a launch
  // with more workgroups than compute units runs as TWO rounds even so (18 v 10 us with three 4-tile Gram problems)
  int tile0[GS_MAXPROB + 1];
  // round 6 (fused, the step's FIRST launch): one thread advances the step counters and leaves Adam's bias corrections
  // in fin[net][1 .. 2] (float64 pow / sqrt like torch: ~2000 cycles that sat on the top launch's critical path)
  float* fin;          // [3][8], or nullptr
  int* adam_step;
  const float* lr_dev;
  float lr_actor, lr_critic, beta1, beta2;
  int fin_mask;
  // ... and gathers the loss's per-sample operands (they hang on the row index: a second memory round trip in the top
  // launch) into the tiled path's layout: scal[k][R] (logp, adv_r, adv_c, target_value_r, target_value_c), actg[R][lda]
  const float* sp[5];
  const float* act;
  float* scal;
  float* actg;
  int ld_act, act_dim, lda;
};
__device__ __forceinline__ int gs_locate(const GSArgs& a, int& tile, int& tiles) {
  const int b = blockIdx.x;
  int y = 0;
#pragma unroll
  for (int q = 1; q < GS_MAXPROB; ++q)
    if (q < a.nprob && b >= a.tile0[q]) y = q;
  int lo = a.tile0[0], hi = a.tile0[1];
#pragma unroll
  for (int q = 1; q < GS_MAXPROB; ++q)
    if (q == y) { lo = a.tile0[q]; hi = a.tile0[q + 1]; }
  tile = b - lo;
  tiles = hi - lo;
  return y;
}
__device__ __forceinline__ GSProb gs_pick(const GSArgs& a, int y) {
  // (selected with scalar compares: a run-time index into the by-value argument would go through scratch memory)
  GSProb p = a.p[0];
#pragma unroll
  for (int q = 1; q < GS_MAXPROB; ++q)
    if (q == y) p = a.p[q];
  return p;
}

#ifndef GS_WAVES
#define GS_WAVES 8
#endif
#ifndef GS_KB
#define GS_KB 16  // columns of a contraction block: 16 (one 16-byte piece per lane) or 32 (two adjacent pieces: a whole
#endif            // 128-byte line per lane group of a row)
#define GS_NQ (GS_KB / 16)
#ifndef GS_PF
#define GS_PF (8 / GS_NQ)  // contraction blocks whose loads are in flight together (per wave)
#endif

// 16-byte load of columns c0 .. c0 + 3 of a row whose leading dimension is a multiple of 4, WITHOUT control flow (a
// branch per load keeps the compiler from issuing a wave's loads together): the address is clamped into the row and
// the value selected afterwards.  Padding columns inside the leading dimension are zero by construction (parameter
// blocks; gathered rows; the skinny kernels and the loss kernel write the padding of what they produce).
__device__ __forceinline__ f32x4 gs_load4(const float* __restrict__ row, int c0, int ld, bool ok) {
  ok = ok && c0 < ld;
  const f32x4 v = *reinterpret_cast<const f32x4*>(row + (ok ? c0 : 0));
  return ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
}

// the same four columns from a row of ANY alignment (four-byte loads; columns beyond `lim` are zero)
__device__ __forceinline__ f32x4 gs_load4u(const float* __restrict__ row, int c0, int lim, bool ok) {
  f32x4 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool okq = ok && c0 + q < lim;
    const float t = row[okq ? c0 + q : 0];
    v[q] = okq ? t : 0.f;
  }
  return v;
}

// the cross-wave sum of the 4 row tiles' accumulators, in wave order; returns tile `t`'s sum for lane `ln`
__device__ __forceinline__ f32x4 gs_sum_waves(const float* red, int t, int ln) {
  f32x4 v = *reinterpret_cast<const f32x4*>(red + ((0 * 4 + t) * 64 + ln) * 4);
#pragma unroll
  for (int w = 1; w < GS_WAVES; ++w) v = v + *reinterpret_cast<const f32x4*>(red + ((w * 4 + t) * 64 + ln) * 4);
  return v;
}

// deterministic sum over the first 256 threads of a workgroup of 256 or 64 GS_WAVES threads (result in thread 0; every
// thread of the workgroup must call): wave sums in wave order
__device__ __forceinline__ float gs_sum256(float v, float* red4) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = osa_wave_sum_dpp(v);
  __syncthreads();
  if (lane == 0 && wave < 4) red4[wave] = v;
  __syncthreads();
  return (red4[0] + red4[1]) + (red4[2] + red4[3]);
}

#define GS_TLD 20  // leading dimension of a [64 rows][16 columns] tile in LDS

// Squared norm of rows n0 .. n0 + 15 of a layer's weight gradient dW = dZ^T H WITHOUT forming it (round 6: no
// weight-gradient launch for the clip norm).  With the tile T = dZ[:, n0 .. n0 + 15] in LDS and G = H H^T (64 x 64,
// written by a Gram problem of the layer's forward launch):
//     sum_{n, k} dW[n][k]^2 = sum_n T[:, n]^T G T[:, n] = sum_{n, c} D[n][c] T[c][n],   D = T^T G  (64 MFMAs),
// the critics' L2 term g = dW + c2 W adds 2 c2 <dW, W> + c2^2 |W|^2 with <dW, W> = sum_{r, n} dZ[r][n] Zlin[r][n]
// (Zlin = H W^T, the forward pass's product before the bias: `cross` is the calling thread's share of that sum) and
// |W|^2 from the forward launch's slots; the bias gradient (column sums of T, + c2 b) is formed directly.  The same
// quantity as gs_wgrad_kernel<0>'s, up to float32 summation order.  Waves 0 .. 3 work.
// (this thread's share of the sum; b4 = the layer's bias n0 + 4 g .. + 3, used by wave 0's lanes j == 0: the bias
// gradient = column sums of T come from one more MFMA chain against a matrix of ones, not from a serial LDS loop)
__device__ __forceinline__ float gs_gram_norm(const float* sT, const float (&gv)[16], const float cross, const f32x4 b4,
                                              const int n0, const int out, const float c2) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  float tot = 0.f;
  if (wave < 4) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, ones = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float av = sT[(16 * (s >> 2) + 4 * g + (s & 3)) * GS_TLD + j];  // (row r(s, g) of gs_gram_load)
      acc = OSA_MFMA(av, gv[s], acc);
      if (wave == 0) ones = OSA_MFMA(av, 1.f, ones);  // wave-uniform
    }
    // D[n = 4 g + i][c = 16 wave + j] = acc[i];  ones[i] = sum_r T[r][4 g + i] in every column j
    const f32x4 tv = *reinterpret_cast<const f32x4*>(sT + (16 * wave + j) * GS_TLD + 4 * g);
    tot = (acc[0] * tv[0] + acc[1] * tv[1]) + (acc[2] * tv[2] + acc[3] * tv[3]);
    tot += 2.f * c2 * cross;
    if (wave == 0 && j == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (n0 + 4 * g + i < out) {
          const float gb = ones[i] + c2 * b4[i];
          tot += gb * gb;
        }
    }
  }
  return tot;
}
// the Gram operand of gs_gram_norm for this lane, waves 0 .. 3.  MFMA step s of lane group g contracts row
// r(s, g) = 16 (s >> 2) + 4 g + (s & 3) (any permutation serves as long as both operands use it): the four steps of a
// quarter then need G[r .. r + 3][c], which by symmetry is the 16-byte piece G[c][r .. r + 3] -- four loads instead of 16
__device__ __forceinline__ void gs_gram_load(float (&gv)[16], const float* __restrict__ G, const int R) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int c = 16 * wave + j;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 16 * q + 4 * g;
    const bool ok = G != nullptr && wave < 4 && c < R;
    const f32x4 v = *reinterpret_cast<const f32x4*>(G + (ok ? c * 64 + r : 0));
#pragma unroll
    for (int i = 0; i < 4; ++i) gv[4 * q + i] = ok && r + i < R ? v[i] : 0.f;
  }
}

// grid (sum of the problems' tiles: GSArgs.tile0), 64 GS_WAVES threads.   XI: the rows come from the caller's array
// (GSArgs.xrows)
template <bool XI>
__global__ __launch_bounds__(64 * GS_WAVES) void gs_fwd_kernel(GSArgs a) {
  __shared__ __attribute__((aligned(16))) float red[GS_WAVES * 4 * 64 * 4];
  __shared__ __attribute__((aligned(16))) float gs_stage[XI ? 4 : GS_WAVES * 80 * 32];
  __shared__ float wred[GS_WAVES + 1];
  int tile, tiles;
  const GSProb p = gs_pick(a, gs_locate(a, tile, tiles));
  const int n0 = tile * 16;
  if (XI && a.fin && blockIdx.x == a.tile0[GS_MAXPROB]) {  // the auxiliary workgroup behind the last tile
    const int t = threadIdx.x;
    if (t < a.R) {
      const long row = a.xidx ? a.xidx[t] : t;
#pragma unroll
      for (int q = 0; q < 5; ++q) a.scal[q * a.R + t] = a.sp[q] ? a.sp[q][row] : 0.f;
      if (a.act)
        for (int d = 0; d < a.lda; ++d) a.actg[t * a.lda + d] = d < a.act_dim ? a.act[row * a.ld_act + d] : 0.f;
    }
    if (t >= 64 && t < 67 && ((a.fin_mask >> (t - 64)) & 1)) {
      const int net = t - 64;
      const int step = a.adam_step[net] + 1;
      const float lr = a.lr_dev ? a.lr_dev[net ? 1 : 0] : (net ? a.lr_critic : a.lr_actor);
      a.fin[net * 8 + 1] = (float)((double)lr / (1.0 - pow((double)a.beta1, (double)step)));
      a.fin[net * 8 + 2] = (float)(1.0 / sqrt(1.0 - pow((double)a.beta2, (double)step)));
      a.adam_step[net] = step;
    }
    return;
  }
  if (n0 >= p.ldy) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int R = a.R, K = p.K;
  // contraction blocks of GS_KB columns: lane group g holds columns GS_KB b + GS_NQ 4 g .. of a block as GS_NQ 16-byte
  // pieces; MFMA step (q, s) contracts column GS_KB b + 4 GS_NQ g + 4 q + s -- the same permutation for both operands
  const int nblk = (K + GS_KB - 1) / GS_KB, per = (nblk + GS_WAVES - 1) / GS_WAVES;
  const int b0 = wave * per, b1 = b0 + per < nblk ? b0 + per : nblk;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool wok = n0 + j < p.N;
  long wr = wok ? n0 + j : 0;
  if (XI && p.wrows && a.xidx) wr = a.xidx[wr];
  const float* __restrict__ wrow = p.W + wr * p.ldw;
  const float* __restrict__ xrow[4];
  bool xok[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    xok[t] = 16 * t + j < R;
    long r = xok[t] ? 16 * t + j : 0;
    if (XI && a.xidx) r = a.xidx[r];
    xrow[t] = p.X + r * p.ldx;
  }
  f32x4 wsq4 = {0.f, 0.f, 0.f, 0.f};  // squares of the tile's weights (this lane's share)
  // the epilogue's operands are requested NOW (a load behind the contraction is a memory round trip of its own):
  // bias of the thread's four output columns, the fused top layer's weights of those columns, the tile's 16 biases
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, w2pre[4];
  float bsq = 0.f;
  {
    const int n = n0 + 4 * g;
    if (tid < 256 && p.bias && n < p.ldy) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      w2pre[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (tid < 256 && p.oslab && o < p.N2 && n < p.ldy) w2pre[o] = *reinterpret_cast<const f32x4*>(p.W2 + (long)o * p.ldw2 + n);
    }
    if (p.slot && p.bias && tid < 16 && n0 + tid < p.N) bsq = p.bias[n0 + tid];
  }
#if !defined(GS_NO_STAGE) && GS_KB == 16 && GS_PF == 8
  if constexpr (!XI) {
    // Round 6: the operands reach their MFMA lanes through LDS.  An MFMA operand register holds one ROW per lane (lane =
    // 16 g + j: row j): loaded straight from memory, the 64 lanes of an instruction touch 64 separate 16-byte pieces in
    // 16 rows and the texture unit spends ~65 cycles per instruction on them (tools/skinny_probe.hip: one workgroup ALONE
    // takes as long as the whole grid).  Staged, an instruction covers 8 rows x 128 contiguous bytes (eight lanes per
    // line), each wave writes its sub-chunk of [64 + 16 rows] x [32 columns] to its own LDS buffer and reads the
    // fragments back in operand layout (XOR-swizzled 16-byte pieces: no bank conflicts either way).  Same operand
    // values in the same MFMA order as the direct form: bit-identical.
    float* __restrict__ stg = gs_stage + wave * (80 * 32);
    const int lrow = lane >> 3, lp = lane & 7;
    for (int bb = b0; bb < b1; bb += 8) {
      f32x4 xr[4][8], wr[4][2];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = 16 * (bb + 2 * c) + 4 * (lp ^ lrow);  // (rows 8 i + lrow: (row & 7) == lrow)
        const bool cok = bb + 2 * c + ((lp ^ lrow) >> 2) < b1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int n = n0 + 8 * i + lrow;
          wr[c][i] = gs_load4(p.W + (long)(n < p.N ? n : 0) * p.ldw, col, p.ldw, cok && n < p.N);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = 8 * i + lrow;
          xr[c][i] = gs_load4(p.X + (long)(r < R ? r : 0) * p.ldx, col, p.ldx, cok && r < R);
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (bb + 2 * c < b1) {  // wave-uniform
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(stg + (8 * i + lrow) * 32 + 4 * lp) = xr[c][i];
#pragma unroll
          for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(stg + (64 + 8 * i + lrow) * 32 + 4 * lp) = wr[c][i];
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (bb + 2 * c + u < b1) {  // wave-uniform
              const int pc = 4 * (((4 * u + g) ^ j) & 7);
              const f32x4 wf = *reinterpret_cast<const f32x4*>(stg + (64 + j) * 32 + pc);
              f32x4 xf[4];
#pragma unroll
              for (int t = 0; t < 4; ++t) xf[t] = *reinterpret_cast<const f32x4*>(stg + (16 * t + j) * 32 + pc);
#pragma unroll
              for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = OSA_MFMA(wf[s], xf[t][s], acc[t]);
              wsq4 = wsq4 + wf * wf;
            }
          }
        }
      }
    }
  }
#endif
  // (XI: layer 0 -- obs_dim columns, a block or two per wave: no window of clamped loads)
  constexpr int PFW = XI ? 1 : GS_PF;
#if !defined(GS_NO_STAGE) && GS_KB == 16 && GS_PF == 8
  if constexpr (XI)
#endif
  for (int bb = b0; bb < b1; bb += PFW) {
    // every load of the window is issued before the first MFMA: the weights come from HBM, the rows from L2
    f32x4 wf[PFW][GS_NQ], xf[PFW][GS_NQ][4];
#pragma unroll
    for (int u = 0; u < PFW; ++u)
#pragma unroll
      for (int q = 0; q < GS_NQ; ++q) {
        const int c = GS_KB * (bb + u) + 4 * GS_NQ * g + 4 * q;
        wf[u][q] = (XI && p.wrows) ? gs_load4u(wrow, c, K, wok && bb + u < b1) : gs_load4(wrow, c, p.ldw, wok && bb + u < b1);
      }
#pragma unroll
    for (int u = 0; u < PFW; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < GS_NQ; ++q)
          xf[u][q][t] = XI ? gs_load4u(xrow[t], GS_KB * (bb + u) + 4 * GS_NQ * g + 4 * q, K, xok[t] && bb + u < b1)
                           : gs_load4(xrow[t], GS_KB * (bb + u) + 4 * GS_NQ * g + 4 * q, p.ldx, xok[t] && bb + u < b1);
#pragma unroll
    for (int u = 0; u < PFW; ++u) {
      if (bb + u < b1) {  // wave-uniform
#pragma unroll
        for (int q = 0; q < GS_NQ; ++q) {
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = OSA_MFMA(wf[u][q][s], xf[u][q][t][s], acc[t]);
          wsq4 = wsq4 + wf[u][q] * wf[u][q];
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(red + ((wave * 4 + t) * 64 + lane) * 4) = acc[t];
  if (p.slot) {  // block-uniform
    const float ws_ = osa_wave_sum_dpp((wsq4.x + wsq4.y) + (wsq4.z + wsq4.w));
    if (lane == 0) wred[wave] = ws_;
    if (wave == 0) {
      const float bq = osa_wave_sum_dpp(bsq * bsq);
      if (lane == 0) wred[GS_WAVES] = bq;
    }
  }
  __syncthreads();
  if (p.slot && tid == 0) {
    float wq = 0.f;
#pragma unroll
    for (int w = 0; w < GS_WAVES; ++w) wq += wred[w];
    *reinterpret_cast<f32x4*>(p.slot + 4 * tile) = (f32x4){0.f, wred[GS_WAVES], wq, 0.f};
  }
  if (tid < 256) {
    // D[m = 4 g + r][n = j] of row tile t: output columns n0 + 4 g + r of row 16 t + j
    const int t = tid >> 6, row = 16 * t + j, n = n0 + 4 * g;
    f32x4 v = gs_sum_waves(red, t, lane);
    const bool ok = row < R && n < p.ldy;
    f32x4 lin = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float y = 0.f;  // (padding columns of the row: zero, the next layer's 16-byte loads run over them)
      if (ok && n + r < p.N) {
        y = v[r];
        lin[r] = y;
        y += bias4[r];
        if (p.act >= 0) y = gm_act(y, p.act);
      }
      v[r] = y;
    }
    if (ok) {
      *reinterpret_cast<f32x4*>(p.Y + (long)row * p.ldy + n) = v;
      if (p.Ylin) *reinterpret_cast<f32x4*>(p.Ylin + (long)row * p.ldy + n) = lin;
    }
    if (p.oslab) {  // block-uniform: the top layer's partial outputs from this tile's 16 columns (sum over g by shuffles)
      for (int o = 0; o < p.ldo2; ++o) {
        float part = 0.f;
        if (ok && o < p.N2) {
          f32x4 w2 = w2pre[0];
          if (o == 1) w2 = w2pre[1];
          if (o == 2) w2 = w2pre[2];
          if (o == 3) w2 = w2pre[3];
          if (o >= 4) w2 = *reinterpret_cast<const f32x4*>(p.W2 + (long)o * p.ldw2 + n);
          part = (v[0] * w2[0] + v[1] * w2[1]) + (v[2] * w2[2] + v[3] * w2[3]);
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (g == 0 && row < R) p.oslab[((long)tile * 64 + row) * p.ldo2 + o] = part;
      }
    }
  }
}

// grid (sum of the problems' tiles), 64 GS_WAVES threads
__global__ __launch_bounds__(64 * GS_WAVES) void gs_bwd_kernel(GSArgs a) {
  __shared__ __attribute__((aligned(16))) float red[GS_WAVES * 4 * 64 * 4];
  __shared__ __attribute__((aligned(16))) float sT[64 * GS_TLD];
  __shared__ __attribute__((aligned(16))) float gs_stage[GS_WAVES * 64 * 32];
  __shared__ __attribute__((aligned(16))) float gs_stage_w[GS_WAVES * 32 * GS_TLD];
  __shared__ float red4[4];
  int tile, tiles;
  const GSProb p = gs_pick(a, gs_locate(a, tile, tiles));
  // Column tiles 2 m and 2 m + 1 read the two 64-byte halves of the SAME 128-byte lines of every weight row: in dispatch
  // order they land on different XCCs (workgroup i -> XCC i mod 8) and both L2s fetch the line -- the launch then moves
  // 34 MB instead of 21 (profiles/r5_pmc_traffic_general_1024_B64_table.md).  Remapped, the pair shares an XCC.
  int bx = tile;
#ifndef GS_BWD_NO_PAIR
  if ((tiles & 15) == 0 && ((blockIdx.x - tile) & 7) == 0) {
    const int xcd = bx & 7, slot = bx >> 3;
    bx = 2 * ((slot >> 1) * 8 + xcd) + (slot & 1);
  }
#endif
  const int k0 = bx * 16;
  if (k0 >= p.ldy) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int R = a.R, N = p.N;
  const int nblk = (N + 15) / 16, per = (nblk + GS_WAVES - 1) / GS_WAVES;
  const int b0 = wave * per, b1 = b0 + per < nblk ? b0 + per : nblk;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool kok = k0 + j < p.K;
  const float* __restrict__ wcol = p.W + (kok ? k0 + j : 0);
  const float* __restrict__ xrow[4];
  bool xok[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    xok[t] = 16 * t + j < R;
    xrow[t] = p.X + (long)(xok[t] ? 16 * t + j : 0) * p.ldx;
  }
  float gv[16];  // (round 6) this lane's share of the Gram matrix for the norm of the produced rows' weight gradient
  const bool norm = a.fused && p.nslot;
  if (norm) gs_gram_load(gv, p.G, R);
  // the epilogue's operands are requested now (see gs_fwd_kernel)
  f32x4 hpre = {0.f, 0.f, 0.f, 0.f}, zpre = {0.f, 0.f, 0.f, 0.f};
  f32x4 bpre = {0.f, 0.f, 0.f, 0.f};
  if (norm && wave == 0 && j == 0 && k0 + 4 * g < p.ldy) bpre = *reinterpret_cast<const f32x4*>(p.bvec + k0 + 4 * g);
  if (tid < 256) {
    const int row = 16 * (tid >> 6) + j, k = k0 + 4 * g;
    if (row < R && k < p.ldy) {
      if (p.act >= 0) hpre = *reinterpret_cast<const f32x4*>(p.aux + (long)row * p.ldaux + k);
      if (norm && p.Zlin) zpre = *reinterpret_cast<const f32x4*>(p.Zlin + (long)row * p.ldy + k);
    }
  }
#if !defined(GS_NO_STAGE) && GS_PF == 8
  // (round 6) the dZ rows reach their MFMA lanes through LDS -- coalesced 8 rows x 128 bytes per instruction, see
  // gs_fwd_kernel; the weight fragments are 4-byte loads of 16 consecutive columns per row already
  // ... and so do the weights: W[n][k0 .. k0 + 15] of a block's 16 rows n as one instruction of 16-byte pieces (16 rows x
  // 64 bytes) instead of four 4-byte loads per lane; the fragments a[n = 4 g + s][k0 + j] are read back as scalars from a
  // [32 rows][20] tile (row stride 20: the four lane groups hit disjoint banks)
  float* __restrict__ stg = gs_stage + wave * (64 * 32);
  float* __restrict__ stw = gs_stage_w + wave * (32 * GS_TLD);
  (void)wcol;
  (void)kok;
  const int lrow = lane >> 3, lp = lane & 7;
  const int wrow_ = lane >> 2, wp = lane & 3;
  for (int bb = b0; bb < b1; bb += 8) {
    f32x4 wr[8];
    f32x4 xr[4][8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = 16 * (bb + u) + wrow_;
      const bool ok = bb + u < b1 && n < N && k0 + 4 * wp < p.ldw;
      wr[u] = gs_load4(p.W + (long)(ok ? n : 0) * p.ldw, k0 + 4 * wp, p.ldw, ok);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = 16 * (bb + 2 * c) + 4 * (lp ^ lrow);
      const bool cok = bb + 2 * c + ((lp ^ lrow) >> 2) < b1;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = 8 * i + lrow;
        xr[c][i] = gs_load4(p.X + (long)(r < R ? r : 0) * p.ldx, col, p.ldx, cok && r < R);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (bb + 2 * c < b1) {  // wave-uniform
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(stg + (8 * i + lrow) * 32 + 4 * lp) = xr[c][i];
#pragma unroll
        for (int u = 0; u < 2; ++u) *reinterpret_cast<f32x4*>(stw + (16 * u + wrow_) * GS_TLD + 4 * wp) = wr[2 * c + u];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (bb + 2 * c + u < b1) {  // wave-uniform
            const int pc = 4 * (((4 * u + g) ^ j) & 7);
            f32x4 xf[4];
            float wf[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) wf[s] = stw[(16 * u + 4 * g + s) * GS_TLD + j];
#pragma unroll
            for (int t = 0; t < 4; ++t) xf[t] = *reinterpret_cast<const f32x4*>(stg + (16 * t + j) * 32 + pc);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int t = 0; t < 4; ++t) acc[t] = OSA_MFMA(wf[s], xf[t][s], acc[t]);
          }
        }
      }
    }
  }
#else
  for (int bb = b0; bb < b1; bb += GS_PF) {
    // A fragments: step s of block u contracts n = 16 (bb + u) + 4 g + s (the same permutation for both operands);
    // branch-free (clamped address, value selected), all loads of the window before the first MFMA
    float wf[GS_PF][4];
    f32x4 xf[GS_PF][4];
#pragma unroll
    for (int u = 0; u < GS_PF; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int n = 16 * (bb + u) + 4 * g + s;
        const bool ok = kok && bb + u < b1 && n < N;
        const float wv = wcol[(long)(ok ? n : 0) * p.ldw];
        wf[u][s] = ok ? wv : 0.f;
      }
#pragma unroll
    for (int u = 0; u < GS_PF; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) xf[u][t] = gs_load4(xrow[t], 16 * (bb + u) + 4 * g, p.ldx, xok[t] && bb + u < b1);
#pragma unroll
    for (int u = 0; u < GS_PF; ++u) {
      if (bb + u < b1) {  // wave-uniform
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = OSA_MFMA(wf[u][s], xf[u][t][s], acc[t]);
      }
    }
  }
#endif
#pragma unroll
  for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(red + ((wave * 4 + t) * 64 + lane) * 4) = acc[t];
  __syncthreads();
  float cross = 0.f;
  if (tid < 256) {
    const int t = tid >> 6, row = 16 * t + j, k = k0 + 4 * g;
    f32x4 v = gs_sum_waves(red, t, lane);
    const bool ok = row < R && k < p.ldy;
    if (ok) {
      const f32x4 h = hpre;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float y = 0.f;
        if (k + r < p.K) {
          y = v[r];
          if (p.act >= 0) y *= gm_dact(h[r], p.act);
        }
        v[r] = y;
      }
      *reinterpret_cast<f32x4*>(p.Y + (long)row * p.ldy + k) = v;
    } else {
      v = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (norm) {
      *reinterpret_cast<f32x4*>(sT + row * GS_TLD + 4 * g) = v;
      cross = (v[0] * zpre[0] + v[1] * zpre[1]) + (v[2] * zpre[2] + v[3] * zpre[3]);
    }
  }
  if (norm) {  // block-uniform
    __syncthreads();
    const float gq = gs_sum256(gs_gram_norm(sT, gv, cross, bpre, k0, p.K, p.c2), red4);
    if (tid == 0) *reinterpret_cast<f32x4*>(p.nslot + 4 * bx) = (f32x4){gq, 0.f, 0.f, 0.f};
  }
}

// ---- loss + backward through the TOP layer in one launch (round 5) -----------------------------------------------------
// The top layer's backward-data product contracts over act_dim (or 1) outputs: every workgroup of that launch can afford
// to compute the loss of all <= 64 rows itself (dL/d(output) into LDS) instead of waiting for a 5 us launch that does it
// once.  Workgroup 0 of a network is the writer: dL/d(output) rows for the weight-gradient launch, block partials of
// dL/d(log_std), loss statistics.  Networks without a hidden layer take part with zero output columns (loss only).
#define GS_TOP_LDZ 36  // leading dimension of the dL/d(output) image in LDS (top layers up to 32 wide)
#ifdef GS_CLOCKS  // (tools/skinny_probe.hip: phase stamps of one workgroup)
__device__ long long gs_clk[2][16];
#define GS_STAMP(i) do { if (threadIdx.x == 0 && tile < 2 && p.net == 0) gs_clk[tile][i] = __builtin_readcyclecounter(); } while (0)
#ifdef GS_CLOCKS_WAIT  // every group of loads waited for: what each costs alone
#define GS_WSTAMP(i) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (threadIdx.x == 64 && tile == 1 && p.net == 0) gs_clk[0][8 + i] = __builtin_readcyclecounter(); } while (0)
#else
#define GS_WSTAMP(i) do { } while (0)
#endif
#else
#define GS_WSTAMP(i) do { } while (0)
#define GS_STAMP(i) do { } while (0)
#endif
struct GSTopFin {  // (round 6) what gs_wgrad_kernel<0>'s tail workgroup did: the writer workgroups of the top launch do it
  float* stats;
  float entropy_coef;
  int loss_kind, act_dim;
};
// grid (sum over the problems of max(1, tiles)), 256 threads (wave = row tile)
// a.fused (round 6, mode 0): (1) the top layer's OUTPUT is the sum of the partial slabs the forward launch of the layer
// below left (GSProb.oslab; no launch for a 1024 -> 2 layer), (2) norm slots: the top layer's weight-gradient block of
// this workgroup's 16 input columns directly (dZ is <= 32 wide), the rows k0 .. k0 + 15 of the layer below through its
// Gram matrix (gs_gram_norm), (3) the writer workgroup: log_std, Adam's bias corrections, loss statistics.
__global__ __launch_bounds__(256) void gs_top_kernel(GSArgs a, GLossArgs la, GSTopFin tf) {
  __shared__ __attribute__((aligned(16))) float sDZ[64 * GS_TOP_LDZ];
  __shared__ __attribute__((aligned(16))) float sOut[64 * GS_TOP_LDZ];
  __shared__ __attribute__((aligned(16))) float sT[64 * GS_TLD];
  __shared__ __attribute__((aligned(16))) float sA[64 * GS_TLD];
  __shared__ float sAct[64 * 32];
  __shared__ float sLS[4 * 32];  // the actor's log_std and its tables (GLossLds)
  __shared__ float sLam;         // the Lagrange multiplier
  __shared__ float sBT[8];       // the fused top layer's bias
  __shared__ __attribute__((aligned(16))) float sPart[4 * 64 * 8];
  __shared__ float sums[2 + 32];
  __shared__ float red[4];
  __shared__ float red3[4 * 3];
  int tile, tiles;
  const GSProb p = gs_pick(a, gs_locate(a, tile, tiles));
  const int k0 = tile * 16;
  const bool writer = tile == 0;
  if (!writer && k0 >= p.ldy) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int net = p.net, R = a.R, ldz = la.ldz[net];
  const bool fused = a.fused != 0;
  GS_STAMP(0);
  GS_WSTAMP(0);
  // ---- every operand that depends on nothing is requested before the first dependent step (each load issued later is
  // a memory round trip of its own on the critical path of this latency-bound launch)
  const long lrow = tid < R && (!fused || la.ext_on) ? (la.idx ? la.idx[tid] : tid) : 0;  // (fused: only FOCOPS / P3O's old_mean)
  float gv[16];
  if (fused) gs_gram_load(gv, p.G, R);
  GS_WSTAMP(1);
  const int N = p.N;
  const bool cols = k0 < p.ldy;
  const bool kok = cols && k0 + j < p.K;
  const float* __restrict__ wcol = p.W + (kok ? k0 + j : 0);
  float wf[2][4];
#pragma unroll
  for (int bb = 0; bb < 2; ++bb)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int n = 16 * bb + 4 * g + s;
      const bool ok = kok && n < N;
      const float wv = wcol[(long)(ok ? n : 0) * p.ldw];
      wf[bb][s] = ok ? wv : 0.f;
    }
  GS_WSTAMP(2);
  const int row = 16 * wave + j, k = k0 + 4 * g;
  const bool ok = cols && row < R && k < p.ldy;
  f32x4 h = {0.f, 0.f, 0.f, 0.f}, zl = {0.f, 0.f, 0.f, 0.f}, bpre = {0.f, 0.f, 0.f, 0.f};
  f32x4 btop4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (ok && (p.act >= 0 || fused)) h = *reinterpret_cast<const f32x4*>(p.aux + (long)row * p.ldaux + k);
  if (ok && fused && p.Zlin) zl = *reinterpret_cast<const f32x4*>(p.Zlin + (long)row * p.ldy + k);
  if (fused) {
    if (cols && wave == 0 && j == 0 && k < p.ldy) bpre = *reinterpret_cast<const f32x4*>(p.bvec + k);
    if (writer && wave == 3 && j == 0) {
#pragma unroll
      for (int hb = 0; hb < 2; ++hb)
        if (16 * hb + 4 * g < (N + 3) / 4 * 4) btop4[hb] = *reinterpret_cast<const f32x4*>(p.btop + 16 * hb + 4 * g);
    }
  }
  // the top layer's weights of this workgroup's block in the layout of the norm's MFMA result (waves 1, 2)
  float wdir[4] = {0.f, 0.f, 0.f, 0.f};
  if (fused && (wave == 1 || wave == 2)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = 16 * (wave - 1) + 4 * g + i;
      if (kok && n < N) wdir[i] = p.W[(long)n * p.ldw + k0 + j];
    }
  }
  GS_WSTAMP(3);
  // the fused top layer's partial outputs: lane = row, wave w sums the slabs w, w + 4, ... (16-byte pieces of the row, all
  // of them requested at once), the four partial sums meet in LDS
  const bool ofuse = fused && p.oslab != nullptr && p.ldo2 <= 8 && p.nbo <= 64;
  f32x4 osum[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (ofuse) {
    const long ne = 64 * p.ldo2;
#pragma unroll
    for (int q4 = 0; q4 < 2; ++q4) {
      if (4 * q4 < p.ldo2) {  // block-uniform
        f32x4 t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int b = wave + 4 * i;
          const bool bok = lane < R && b < p.nbo;
          const f32x4 x = *reinterpret_cast<const f32x4*>(p.oslab + (bok ? b * ne + lane * p.ldo2 + 4 * q4 : 0));
          t[i] = bok ? x : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) osum[q4] = osum[q4] + t[i];
      }
    }
  }
  GS_WSTAMP(4);
  // the loss's per-sample operands, gathered by the step's first launch (GSArgs.scal / actg): scalars in registers of
  // wave 0 (lane = row); action rows, log_std with its tables and the Lagrange multiplier in LDS
  float scv[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (fused) {
    if (tid < la.act_dim) {
      const float ls = la.log_std[tid], sd = expf(ls);
      sLS[tid] = ls;
      sLS[32 + tid] = sd;
      sLS[64 + tid] = logf(sd);
      sLS[96 + tid] = 1.f / (sd * sd);
    }
    if (tid == 32) sLam = la.lagrange ? *la.lagrange : 0.f;
    if (ofuse && tid >= 64 && tid < 64 + p.ldo2) sBT[tid - 64] = tid - 64 < N ? p.btop[tid - 64] : 0.f;
    if (tid < R) {
#pragma unroll
      for (int q = 0; q < 5; ++q) scv[q] = la.scal[q * R + tid];
      if (net == 0)
        for (int d = 0; d < la.act_dim; ++d) sAct[tid * 32 + d] = la.actg[tid * la.lda + d];
    }
  }
  GS_WSTAMP(5);
  GS_STAMP(1);
  const float* outp = la.out[net];
  int ldop = la.ldo[net];
  if (ofuse) {  // block-uniform
#pragma unroll
    for (int q4 = 0; q4 < 2; ++q4)
      if (4 * q4 < p.ldo2) *reinterpret_cast<f32x4*>(sPart + (wave * 64 + lane) * 8 + 4 * q4) = osum[q4];
    outp = sOut;
    ldop = GS_TOP_LDZ;
  } else if (fused && p.oslab) {
    const int ne = 64 * p.ldo2;
    for (int e = tid; e < ne; e += 256) {
      const int orow = e / p.ldo2, o = e - orow * p.ldo2;
      float v = 0.f;
      if (orow < R) {
        const float* __restrict__ sl = p.oslab + e;
        for (int b = 0; b < p.nbo; ++b) v += sl[(long)b * ne];
        if (o < N) v += p.btop[o];
        if (writer) const_cast<float*>(la.out[net])[(long)orow * la.ldo[net] + o] = v;
      }
      sOut[orow * GS_TOP_LDZ + o] = v;
    }
    outp = sOut;
    ldop = GS_TOP_LDZ;
  }
  GS_STAMP(2);
  if (fused && !p.oslab) {  // (a top layer wider than 8 outputs ran as a launch of its own: its rows into the LDS image)
    for (int e = tid; e < R * ldop; e += 256) sOut[(e / ldop) * GS_TOP_LDZ + e % ldop] = outp[e];
  }
  for (int e = tid; e < 64 * GS_TOP_LDZ; e += 256) sDZ[e] = 0.f;
  __syncthreads();
  // (rows beyond R: nothing written, the image stays zero)
  if (fused) {
    if (wave == 0) {  // <= 64 rows: wave 0 alone, wave sums
      if (ofuse && tid < R) {  // this row's outputs = bias + the four partial sums, in wave order
        for (int o = 0; o < p.ldo2; ++o) {
          float v = (sPart[(0 * 64 + tid) * 8 + o] + sPart[(1 * 64 + tid) * 8 + o]) +
                    (sPart[(2 * 64 + tid) * 8 + o] + sPart[(3 * 64 + tid) * 8 + o]);
          if (o < N) v += sBT[o];
          sOut[tid * GS_TOP_LDZ + o] = v;
          if (writer) const_cast<float*>(la.out[net])[(long)tid * la.ldo[net] + o] = v;
        }
      }
      GLossLds acc;
#pragma unroll
      for (int q = 0; q < 5; ++q) acc.sc_[q] = scv[q];
      acc.lam_ = sLam;
      acc.act_ = (const lds_f*)sAct + tid * 32;
      acc.out_ = (const lds_f*)sOut + tid * GS_TOP_LDZ;
      acc.ls_ = (const lds_f*)sLS;
      acc.dz_ = (lds_f*)sDZ + tid * GS_TOP_LDZ;
      // (ldzp = the layer's width: the image is zero already, no padding loop)
      gm_loss_body<true>(la, net, tid, 0, red, net == 0 ? la.act_dim : 1, writer, acc, sums, lrow);
    }
  } else {
    const GLossMem acc{la, tid, lrow, outp + (long)tid * ldop, sDZ + tid * GS_TOP_LDZ};
    gm_loss_body<false>(la, net, tid, 0, red, GS_TOP_LDZ, writer, acc, nullptr, lrow);
  }
  __syncthreads();
  GS_STAMP(3);
  if (writer)
    for (int e = tid; e < R * ldz; e += 256) la.dz[net][e] = sDZ[(e / ldz) * GS_TOP_LDZ + e % ldz];
  float tq = 0.f, tp = 0.f;  // the top layer's slot of this workgroup
  if (fused && writer) {
    // ---- the actor's log_std, statistics (gs_wgrad_kernel<0>'s tail; the bias corrections: the step's first launch)
    const bool critic = net != 0;
    if (!critic && tid < tf.act_dim && tf.loss_kind != 2) {
      const float g_ = sums[2 + tid] - tf.entropy_coef / (float)tf.act_dim;
      tq += g_ * g_;
    }
    if (tid == 0) {
      if (tf.stats && tf.loss_kind != 2) {
        const float invB = 1.f / (float)R;
        if (net == 0) {
          float ent = 0.f;
          for (int d = 0; d < tf.act_dim; ++d) ent += 1.41893853320467274178f + sLS[d];
          ent /= (float)tf.act_dim;
          tf.stats[2] = sums[0] * invB - tf.entropy_coef * ent;
          tf.stats[3] = sums[1] * invB;
          tf.stats[4] = ent;
        } else {
          tf.stats[net - 1] = sums[0] * invB;
        }
      }
    }
  }
  if (!cols) return;  // (a writer without columns: never with a.fused, whose networks have a hidden layer)
  GS_STAMP(4);
  // ---- dZ'[r][k] = (sum_n dZ[r][n] W[n][k]) act'(H[r][k]) for the 16 columns k0 .., row tile = wave
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int bb = 0; bb < 2; ++bb) {
    if (16 * bb < N) {  // block-uniform
      const f32x4 xf = *reinterpret_cast<const f32x4*>(sDZ + (16 * wave + j) * GS_TOP_LDZ + 16 * bb + 4 * g);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = OSA_MFMA(wf[bb][s], xf[s], acc);
    }
  }
  if (ok) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float y = 0.f;
      if (k + r < p.K) {
        y = acc[r];
        if (p.act >= 0) y *= gm_dact(h[r], p.act);
      }
      acc[r] = y;
    }
    *reinterpret_cast<f32x4*>(p.Y + (long)row * p.ldy + k) = acc;
  } else {
    acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  if (!fused) return;
  *reinterpret_cast<f32x4*>(sT + row * GS_TLD + 4 * g) = acc;
  *reinterpret_cast<f32x4*>(sA + row * GS_TLD + 4 * g) = h;
  const float cross = (acc[0] * zl[0] + acc[1] * zl[1]) + (acc[2] * zl[2] + acc[3] * zl[3]);
  __syncthreads();
  GS_STAMP(5);
  // ---- norm of the top layer's weight gradient for the input columns k0 .. k0 + 15, formed directly (the layer is at most
  // 32 outputs wide): dW[n][k] = sum_r dZ[r][n] H[r][k] as one MFMA chain per 16 outputs (waves 1, 2); its bias
  // gradient = column sums of dZ against a matrix of ones (writer, wave 3)
  if (wave == 1 || wave == 2) {
    const int hb = wave - 1;
    if (16 * hb < N) {  // wave-uniform
      f32x4 dw = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 16; ++s)
        dw = OSA_MFMA(sDZ[(4 * s + g) * GS_TOP_LDZ + 16 * hb + j], sA[(4 * s + g) * GS_TLD + j], dw);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (kok && 16 * hb + 4 * g + i < N) {  // D[n = 16 hb + 4 g + i][k = k0 + j]
          const float w = wdir[i], gg = dw[i] + p.c2 * w;
          tq += gg * gg;
          if (net != 0) tp += w * w;
        }
    }
  }
  if (writer && wave == 3) {
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
      if (16 * hb < N) {  // block-uniform
        f32x4 ones = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; ++s) ones = OSA_MFMA(sDZ[(4 * s + g) * GS_TOP_LDZ + 16 * hb + j], 1.f, ones);
        if (j == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (16 * hb + 4 * g + i < N) {
              const float bw = btop4[hb][i], gb = ones[i] + p.c2 * bw;
              tq += gb * gb;
              if (net != 0) tp += bw * bw;
            }
        }
      }
  }
  // ---- rows k0 .. k0 + 15 of the layer below through its input rows' Gram matrix
  float gq = gs_gram_norm(sT, gv, cross, bpre, k0, p.K, p.c2);
  GS_STAMP(6);
  tq = osa_wave_sum_dpp(tq);
  tp = osa_wave_sum_dpp(tp);
  gq = osa_wave_sum_dpp(gq);
  if (lane == 0) {
    red3[wave * 3 + 0] = tq;
    red3[wave * 3 + 1] = tp;
    red3[wave * 3 + 2] = gq;
  }
  __syncthreads();
  if (tid == 0) {
    *reinterpret_cast<f32x4*>(p.tslot + 4 * tile) =
        (f32x4){(red3[0] + red3[3]) + (red3[6] + red3[9]), (red3[1] + red3[4]) + (red3[7] + red3[10]), 0.f, 0.f};
    *reinterpret_cast<f32x4*>(p.nslot + 4 * tile) = (f32x4){(red3[2] + red3[5]) + (red3[8] + red3[11]), 0.f, 0.f, 0.f};
  }
  GS_STAMP(7);
}

struct GSWLayer {
  const float* dZ;  // [R][ldz]   dL/d(pre-activation) of this layer
  const float* H;   // [R][ldh]   the layer's input rows
  int ldz, ldh, out, in, ldw, oW, ob, tile0, tk;
  // != 0 (layer 0 without a gather launch): H is the caller's observation array, row r = H[(hidx ? hidx[r] : r) * ldh ..],
  // any alignment (four-byte loads)
  int hrows;
  const long* hidx;
};
struct GSWArgs {
  GSWLayer l[3][GM_MAXL];
  int L[3], ntile[3], oLS[3];
  int R, P, maxT1, act_dim, lda, nblk, nets_mask, loss_kind, write_grads, use_critic_norm;
  // fold != 0 (mode 0): no gm_final_kernel -- phase 0's tail workgroup advances the step counter and leaves Adam's
  // bias corrections in fin, every workgroup of phase 1 sums the norm partials itself (gm_final_kernel's order)
  // fold == 2 (round 6): no phase 0 at all -- the partials are the norm slots of the other launches (gs_gram_norm)
  int fold, use_max_grad_norm;
  const float* slots;  // [3][slot_stride][4]
  int slot_stride, nslot[3];
  float* params;
  float* adam_m;
  float* adam_v;
  float* grads;
  float* npart;       // [3][maxT1][2]
  float* fin;         // [3][8]
  int* adam_step;
  const float* lr_dev;
  const float* dls;   // [nblk][lda]
  const float* lpart; // [3][nblk][4]
  float* stats;
  float entropy_coef, critic_norm_coef, beta1, beta2, eps, max_grad_norm, lr_actor, lr_critic;
};

#define GS_LDT 68  // leading dimension of the [row][64] operand tiles in LDS

// grid (maxT1, 3), 256 threads.  PHASE 0: gradient tile (+ L2 term), squared-norm partials, the gradient itself if
// a.write_grads; PHASE 1: gradient tile again, clip factor from the partials, Adam in place.
template <int PHASE>
__global__ __launch_bounds__(256) void gs_wgrad_kernel(GSWArgs a) {
  __shared__ __attribute__((aligned(16))) float sZ[64 * GS_LDT];
  __shared__ __attribute__((aligned(16))) float sH[64 * GS_LDT];
  __shared__ float red[4];
  const int net = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
  if (!((a.nets_mask >> net) & 1)) return;
  const bool critic = net != 0;
  const bool l2 = critic && a.use_critic_norm;
  const float c2 = 2.f * a.critic_norm_coef;
  float* __restrict__ P_ = a.params + (long)net * a.P;
  float* __restrict__ M_ = a.adam_m + (long)net * a.P;
  float* __restrict__ V_ = a.adam_v + (long)net * a.P;
  float* __restrict__ G_ = a.grads ? a.grads + (long)net * a.P : nullptr;
  if (b > a.ntile[net]) {  // (the grid follows the network with the most tiles)
    if (PHASE == 0 && tid == 0) {
      a.npart[((long)net * a.maxT1 + b) * 2 + 0] = 0.f;
      a.npart[((long)net * a.maxT1 + b) * 2 + 1] = 0.f;
    }
    return;
  }
  const bool tail = b == a.ntile[net];
  // ---- the tile and its parameter rows; the Adam operands are requested BEFORE the operand tiles (two independent
  // memory round trips in flight together)
  int li = 0;
#pragma unroll
  for (int q = 1; q < GM_MAXL; ++q)
    if (q < a.L[net] && b >= a.l[net][q].tile0) li = q;
  // (selected with scalar compares: a run-time index into the by-value argument would go through scratch memory)
  GSWLayer ly = a.l[net][0];
#pragma unroll
  for (int q = 1; q < GM_MAXL; ++q)
    if (q == li) ly = a.l[net][q];
  const int qt = b - ly.tile0, n0 = 64 * (qt / ly.tk), k0 = 64 * (qt % ly.tk);
  const int R = a.R;
  const int lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int kc = k0 + 4 * j;
  f32x4 w4[4], m4[4], v4[4];
  bool rok[4];
  long ro[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + 16 * wave + 4 * g + r;
    rok[r] = !tail && kc < ly.ldw && n < ly.out;
    ro[r] = rok[r] ? ly.oW + (long)n * ly.ldw + kc : 0;  // (0: a readable address, the value is never used)
    w4[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (PHASE == 1 || critic) w4[r] = *reinterpret_cast<const f32x4*>(P_ + ro[r]);
    if (PHASE == 1) {
      // (the moments are read and written once per step and by nobody else: non-temporal)
      m4[r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(M_ + ro[r]));
      v4[r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(V_ + ro[r]));
    }
  }
  float coef = 1.f, step_size = 0.f, ibc2 = 0.f, total_norm = 0.f, total_psq = 0.f;
  if (PHASE == 1) {
    step_size = a.fin[net * 8 + 1];
    ibc2 = a.fin[net * 8 + 2];
    if (a.fold) {
      float gq = 0.f, pq = 0.f;
      if (a.fold == 2) {  // (round 6) the slots of the forward / top / backward launches: {g^2 part, b^2, w^2, -}
        const float cc = l2 ? c2 * c2 : 0.f;
        const f32x4* __restrict__ sl = reinterpret_cast<const f32x4*>(a.slots) + (long)net * a.slot_stride;
        for (int k = tid; k < a.nslot[net]; k += 256) {
          const f32x4 v = sl[k];
          gq += v[0] + cc * v[2];
          pq += v[1] + v[2];
        }
      } else
      for (int k = tid; k < a.maxT1; k += 256) {
        gq += a.npart[((long)net * a.maxT1 + k) * 2 + 0];
        pq += a.npart[((long)net * a.maxT1 + k) * 2 + 1];
      }
      gq = gm_block_sum(gq, red);
      total_psq = gm_block_sum(pq, red);
      total_norm = sqrtf(gq);
      if (a.use_max_grad_norm) {
        coef = a.max_grad_norm / (total_norm + 1e-6f);
        coef = coef > 1.f ? 1.f : coef;
      }
    } else {
      coef = a.fin[net * 8 + 0];
    }
  }
  float gsq = 0.f, psq = 0.f;
  if (tail) {
    // ---- tail of the block: the actor's log_std (block partials of gm_loss_kernel - entropy term), zero padding
    const int e = a.oLS[net] + tid;
    const bool ls = !critic && tid < a.act_dim && a.loss_kind != 2;
    float g_ = 0.f;
    if (ls) {
      for (int k = 0; k < a.nblk; ++k) g_ += a.dls[(long)k * a.lda + tid];
      g_ -= a.entropy_coef / (float)a.act_dim;
    }
    if (PHASE == 0) {
      if (a.write_grads)
        for (int q = e; q < a.P; q += 256) G_[q] = (q == e) ? g_ : 0.f;
      gsq = gm_block_sum(g_ * g_, red);
      if (tid == 0) {
        a.npart[((long)net * a.maxT1 + b) * 2 + 0] = gsq;
        a.npart[((long)net * a.maxT1 + b) * 2 + 1] = 0.f;
        if (a.fold) {  // Adam's bias corrections of THIS step (float64 like torch), step counter advanced
          const int step = a.adam_step[net] + 1;
          const float lr = a.lr_dev ? a.lr_dev[critic ? 1 : 0] : (critic ? a.lr_critic : a.lr_actor);
          a.fin[net * 8 + 1] = (float)((double)lr / (1.0 - pow((double)a.beta1, (double)step)));
          a.fin[net * 8 + 2] = (float)(1.0 / sqrt(1.0 - pow((double)a.beta2, (double)step)));
          a.adam_step[net] = step;
        }
      }
      if (a.stats && a.loss_kind != 2) {  // loss statistics of the step (as gm_reduce_kernel's block 0)
        float l = 0.f, r = 0.f;
        for (int k = tid; k < a.nblk; k += 256) {
          l += a.lpart[((long)net * a.nblk + k) * 4 + 0];
          r += a.lpart[((long)net * a.nblk + k) * 4 + 1];
        }
        l = gm_block_sum(l, red);
        r = gm_block_sum(r, red);
        if (tid == 0) {
          const float invB = 1.f / (float)a.R;
          if (net == 0) {
            float ent = 0.f;
            for (int d = 0; d < a.act_dim; ++d) ent += 1.41893853320467274178f + P_[a.oLS[0] + d];
            ent /= (float)a.act_dim;
            a.stats[2] = l * invB - a.entropy_coef * ent;
            a.stats[3] = r * invB;
            a.stats[4] = ent;
          } else {
            a.stats[net - 1] = l * invB;
          }
        }
      }
    } else {
      if (ls) {
        float m = M_[e], v = V_[e];
        P_[e] = osa_adam_update(g_ * coef, m, v, P_[e], a.beta1, a.beta2, step_size, ibc2, a.eps);
        M_[e] = m;
        V_[e] = v;
      }
      if (a.fold && tid == 0 && a.stats) {  // (gm_final_kernel's share of the statistics)
        a.stats[7 + net] = total_norm;
        if (critic) a.stats[4 + net] = total_psq;
      }
    }
    return;
  }
  // ---- operand tiles: sZ[r][c] = dZ[r][n0 + c], sH[r][c] = H[r][k0 + c] (rows beyond R: zero; columns beyond the
  // layer: the rows' zero padding, or nothing at all past the leading dimension)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int u = tid + 256 * q, r = u >> 4, c4 = 4 * (u & 15);
    *reinterpret_cast<f32x4*>(sZ + r * GS_LDT + c4) = gs_load4(ly.dZ + (long)(r < R ? r : 0) * ly.ldz, n0 + c4, ly.ldz, r < R);
    if (ly.hrows) {  // block-uniform
      long hr = r < R ? r : 0;
      if (ly.hidx) hr = ly.hidx[hr];
      *reinterpret_cast<f32x4*>(sH + r * GS_LDT + c4) = gs_load4u(ly.H + hr * ly.ldh, k0 + c4, ly.in, r < R);
    } else {
      *reinterpret_cast<f32x4*>(sH + r * GS_LDT + c4) = gs_load4(ly.H + (long)(r < R ? r : 0) * ly.ldh, k0 + c4, ly.ldh, r < R);
    }
  }
  __syncthreads();
  // column tile c of the 64 columns holds the columns k0 + 4 j + c: the four accumulators of a lane then are four
  // CONSECUTIVE parameters of a row -- 16-byte accesses to weights, moments and gradient
  f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const float av = sZ[(4 * s + g) * GS_LDT + 16 * wave + j];
    const f32x4 bv = *reinterpret_cast<const f32x4*>(sH + (4 * s + g) * GS_LDT + 4 * j);
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = OSA_MFMA(av, bv[c], acc[c]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (rok[r]) {
      f32x4 gv = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
      const f32x4 w = w4[r];
      if (l2) gv = gv + w * c2;
      if (PHASE == 0) {
        if (critic) psq += (w.x * w.x + w.y * w.y) + (w.z * w.z + w.w * w.w);
        gsq += (gv.x * gv.x + gv.y * gv.y) + (gv.z * gv.z + gv.w * gv.w);
        if (a.write_grads) *reinterpret_cast<f32x4*>(G_ + ro[r]) = gv;
      } else {
        f32x4 m = m4[r], v = v4[r];
        const f32x4 wn = osa_adam_update4(gv * coef, m, v, w, a.beta1, a.beta2, step_size, ibc2, a.eps);
        *reinterpret_cast<f32x4*>(P_ + ro[r]) = wn;
        __builtin_nontemporal_store(m, reinterpret_cast<f32x4*>(M_ + ro[r]));
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(V_ + ro[r]));
      }
    }
  }
  // ---- bias gradient of the tile's 64 outputs (column sums of dZ), by the workgroups of the first column tile
  if (k0 == 0 && tid < 64) {
    const int n = n0 + tid, npad = (ly.out + 3) / 4 * 4;
    if (n < npad) {
      float gb = 0.f;
      for (int r = 0; r < R; ++r) gb += sZ[r * GS_LDT + tid];  // (zero for the padding entries n >= out)
      const long o = ly.ob + n;
      if (n < ly.out) {
        const float w = P_[o];
        if (critic) {
          if (l2) gb += c2 * w;
          if (PHASE == 0) psq += w * w;
        }
        if (PHASE == 0) {
          gsq += gb * gb;
        } else {
          float m = M_[o], v = V_[o];
          P_[o] = osa_adam_update(gb * coef, m, v, w, a.beta1, a.beta2, step_size, ibc2, a.eps);
          M_[o] = m;
          V_[o] = v;
        }
      }
      if (PHASE == 0 && a.write_grads) G_[o] = gb;
    }
  }
  if (PHASE == 0) {
    gsq = gm_block_sum(gsq, red);
    psq = gm_block_sum(psq, red);
    if (tid == 0) {
      a.npart[((long)net * a.maxT1 + b) * 2 + 0] = gsq;
      a.npart[((long)net * a.maxT1 + b) * 2 + 1] = psq;
    }
  }
}

}  // namespace
