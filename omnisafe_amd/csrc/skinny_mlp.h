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
__device__ __forceinline__ void gm_loss_body(const GLossArgs& a, const int net, const long b, const int blk, float* red,
                                             float* __restrict__ dzp, const int ldzp, const bool writer) {
  const bool valid = b < a.R;
  const long row = valid ? (a.idx ? a.idx[b] : b) : 0;  // the sample's row in the caller's arrays
  auto SC = [&](int k) -> float { return a.direct ? a.sp[k][row] : a.scal[(long)k * a.R + b]; };
  auto ACT = [&](int d) -> float { return a.direct ? a.act[row * a.ld_act + d] : a.actg[b * a.lda + d]; };
  const float invB = 1.f / (float)a.R;
  float loss = 0.f, ratio_s = 0.f;
  if (net != 0) {
    if (valid) {
      const float diff = a.out[net][b * a.ldo[net]] - SC(net == 1 ? 3 : 4);
      loss = diff * diff;
      dzp[b * ldzp] = 2.f * diff * invB;
      for (int d = 1; d < ldzp; ++d) dzp[b * ldzp + d] = 0.f;  // (row padding: the skinny kernels' 16-byte loads)
    }
  } else if (a.loss_kind == 2) {
    if (valid)
      for (int d = 0; d < a.act_dim; ++d) {
        const float sd = expf(a.log_std[d]);
        dzp[b * ldzp + d] = a.tmean[b * a.ldt + d] / (sd * sd) * a.fvp_scale;
      }
    if (valid)
      for (int d = a.act_dim; d < ldzp; ++d) dzp[b * ldzp + d] = 0.f;
  } else {
    const float lam = a.lagrange ? *a.lagrange : 0.f;
    float lp = 0.f;
    if (valid)
      for (int d = 0; d < a.act_dim; ++d) {
        const float sd = expf(a.log_std[d]);
        const float z = ACT(d) - a.out[0][b * a.ldo[0] + d];
        lp += -(z * z) / (2.f * (sd * sd)) - logf(sd) - 0.91893853320467274178f;
      }
    const float ratio = valid ? expf(lp - SC(0)) : 0.f;
    // ---- extended surrogates: per-sample KL(pi_theta || pi_old) (torch.distributions.kl._kl_normal_normal), FOCOPS'
    // trust mask with the reference's broadcast semantics (focops.py:84-88: the surrogate term sees the minibatch MEAN
    // of the mask), P3O's kappa * relu(mean(ratio * A_c) + excess) -- the arithmetic of osa_mb_grad_kernel's EXT form.
    // The mask mean and the penalty are minibatch-level: ONE block (the entry point refuses more than 256 rows).
    float kl = 0.f, mask = 1.f, mask_mean = 1.f, cost_w = 0.f;
    const long orow = row;
    if (a.ext_on) {
      if (valid)
        for (int d = 0; d < a.act_dim; ++d) {
          const float ls = a.log_std[d], ls0 = a.old_log_std[d], dl = ls - ls0;
          const float q = expf(dl), isd0 = expf(-ls0);
          const float u = (a.out[0][b * a.ldo[0] + d] - a.old_mean[orow * a.ld_old_mean + d]) * isd0;
          kl += 0.5f * (q * q + u * u - 1.f - 2.f * dl);
        }
      if (a.ext_mask_eta >= 0.f || a.ext_cost_kappa > 0.f) {  // block-uniform
        mask = (a.ext_mask_eta < 0.f || (valid && kl <= a.ext_mask_eta)) ? 1.f : 0.f;
        const float tm = gm_block_sum(valid ? mask : 0.f, red);
        const float tc = gm_block_sum(valid ? ratio * SC(2) : 0.f, red);
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
    // d logp / d mu = z / var;  d logp / d log_std = z^2 / var - 1; block sums of the latter, dimension by dimension
    for (int d = 0; d < a.act_dim; ++d) {
      float dl = 0.f;
      if (valid) {
        const float sd = expf(a.log_std[d]);
        const float iv = 1.f / (sd * sd);
        const float mu = a.out[0][b * a.ldo[0] + d];
        const float z = ACT(d) - mu;
        float dmu = dlogp * z * iv;
        dl = dlogp * (z * z * iv - 1.f);
        if (a.ext_on) {  // d KL / d mu = (mu - mu0) / var0;  d KL / d log_std = var / var0 - 1
          const float ls0 = a.old_log_std[d], q = expf(a.log_std[d] - ls0), isd0 = expf(-ls0);
          const float u = (mu - a.old_mean[orow * a.ld_old_mean + d]) * isd0;
          dmu += dklw * (u * isd0);
          dl += dklw * (q * q - 1.f);
        }
        dzp[b * ldzp + d] = dmu;
      }
      dl = gm_block_sum(dl, red);
      if (threadIdx.x == 0 && writer) a.dls[(long)blk * a.lda + d] = dl;
    }
    if (valid)
      for (int d = a.act_dim; d < ldzp; ++d) dzp[b * ldzp + d] = 0.f;  // (row padding)
  }
  loss = gm_block_sum(loss, red);
  ratio_s = gm_block_sum(ratio_s, red);
  if (threadIdx.x == 0 && writer) {
    float* lp_ = a.lpart + ((long)net * a.nblk + blk) * 4;
    lp_[0] = loss;
    lp_[1] = ratio_s;
  }
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
};
struct GSArgs {
  GSProb p[3];
  int nprob, R;
  // != 0 (forward of layer 0, round 5: no gather launch): X is the CALLER's observation array -- any row stride, any
  // alignment -- and row r of the minibatch is X[(xidx ? xidx[r] : r) * ldx ..]: four-byte loads, K = obs_dim is small
  int xrows;
  const long* xidx;
};

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

// grid (ceil(maxN / 16), nprob), 64 GS_WAVES threads.   XI: the rows come from the caller's array (GSArgs.xrows)
template <bool XI>
__global__ __launch_bounds__(64 * GS_WAVES) void gs_fwd_kernel(GSArgs a) {
  __shared__ __attribute__((aligned(16))) float red[GS_WAVES * 4 * 64 * 4];
  const GSProb p = blockIdx.y == 0 ? a.p[0] : (blockIdx.y == 1 ? a.p[1] : a.p[2]);
  const int n0 = blockIdx.x * 16;
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
  const float* __restrict__ wrow = p.W + (long)(wok ? n0 + j : 0) * p.ldw;
  const float* __restrict__ xrow[4];
  bool xok[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    xok[t] = 16 * t + j < R;
    long r = xok[t] ? 16 * t + j : 0;
    if (XI && a.xidx) r = a.xidx[r];
    xrow[t] = p.X + r * p.ldx;
  }
  // (XI: layer 0 -- obs_dim columns, a block or two per wave: no window of clamped loads)
  constexpr int PFW = XI ? 1 : GS_PF;
  for (int bb = b0; bb < b1; bb += PFW) {
    // every load of the window is issued before the first MFMA: the weights come from HBM, the rows from L2
    f32x4 wf[PFW][GS_NQ], xf[PFW][GS_NQ][4];
#pragma unroll
    for (int u = 0; u < PFW; ++u)
#pragma unroll
      for (int q = 0; q < GS_NQ; ++q)
        wf[u][q] = gs_load4(wrow, GS_KB * (bb + u) + 4 * GS_NQ * g + 4 * q, p.ldw, wok && bb + u < b1);
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
        for (int q = 0; q < GS_NQ; ++q)
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = OSA_MFMA(wf[u][q][s], xf[u][q][t][s], acc[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(red + ((wave * 4 + t) * 64 + lane) * 4) = acc[t];
  __syncthreads();
  if (tid < 256) {
    // D[m = 4 g + r][n = j] of row tile t: output columns n0 + 4 g + r of row 16 t + j
    const int t = tid >> 6, row = 16 * t + j, n = n0 + 4 * g;
    f32x4 v = gs_sum_waves(red, t, lane);
    if (row < R && n < p.ldy) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float y = 0.f;  // (padding columns of the row: zero, the next layer's 16-byte loads run over them)
        if (n + r < p.N) {
          y = v[r];
          if (p.bias) y += p.bias[n + r];
          if (p.act >= 0) y = gm_act(y, p.act);
        }
        v[r] = y;
      }
      *reinterpret_cast<f32x4*>(p.Y + (long)row * p.ldy + n) = v;
    }
  }
}

// grid (ceil(maxK / 16), nprob), 64 GS_WAVES threads
__global__ __launch_bounds__(64 * GS_WAVES) void gs_bwd_kernel(GSArgs a) {
  __shared__ __attribute__((aligned(16))) float red[GS_WAVES * 4 * 64 * 4];
  const GSProb p = blockIdx.y == 0 ? a.p[0] : (blockIdx.y == 1 ? a.p[1] : a.p[2]);
  // Column tiles 2 m and 2 m + 1 read the two 64-byte halves of the SAME 128-byte lines of every weight row: in dispatch
  // order they land on different XCCs (workgroup i -> XCC i mod 8) and both L2s fetch the line -- the launch then moves
  // 34 MB instead of 21 (profiles/r5_pmc_traffic_general_1024_B64_table.md).  Remapped, the pair shares an XCC.
  int bx = blockIdx.x;
#ifndef GS_BWD_NO_PAIR
  if ((gridDim.x & 15) == 0) {
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
#pragma unroll
  for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(red + ((wave * 4 + t) * 64 + lane) * 4) = acc[t];
  __syncthreads();
  if (tid < 256) {
    const int t = tid >> 6, row = 16 * t + j, k = k0 + 4 * g;
    f32x4 v = gs_sum_waves(red, t, lane);
    if (row < R && k < p.ldy) {
      f32x4 h = {0.f, 0.f, 0.f, 0.f};
      if (p.act >= 0) h = *reinterpret_cast<const f32x4*>(p.aux + (long)row * p.ldaux + k);
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
    }
  }
}

// ---- loss + backward through the TOP layer in one launch (round 5) -----------------------------------------------------
// The top layer's backward-data product contracts over act_dim (or 1) outputs: every workgroup of that launch can afford
// to compute the loss of all <= 64 rows itself (dL/d(output) into LDS) instead of waiting for a 5 us launch that does it
// once.  Workgroup 0 of a network is the writer: dL/d(output) rows for the weight-gradient launch, block partials of
// dL/d(log_std), loss statistics.  Networks without a hidden layer take part with zero output columns (loss only).
#define GS_TOP_LDZ 36  // leading dimension of the dL/d(output) image in LDS (top layers up to 32 wide)
// grid (max(1, ceil(maxK / 16)), nprob), 256 threads (wave = row tile)
__global__ __launch_bounds__(256) void gs_top_kernel(GSArgs a, GLossArgs la) {
  __shared__ __attribute__((aligned(16))) float sDZ[64 * GS_TOP_LDZ];
  __shared__ float red[4];
  const GSProb p = blockIdx.y == 0 ? a.p[0] : (blockIdx.y == 1 ? a.p[1] : a.p[2]);
  const int k0 = blockIdx.x * 16;
  const bool writer = blockIdx.x == 0;
  if (!writer && k0 >= p.ldy) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int net = p.net, R = a.R, ldz = la.ldz[net];
  for (int e = tid; e < 64 * GS_TOP_LDZ; e += 256) sDZ[e] = 0.f;
  __syncthreads();
  gm_loss_body(la, net, tid, 0, red, sDZ, GS_TOP_LDZ, writer);  // (rows beyond R: nothing written, the image stays zero)
  __syncthreads();
  if (writer)
    for (int e = tid; e < R * ldz; e += 256) la.dz[net][e] = sDZ[(e / ldz) * GS_TOP_LDZ + e % ldz];
  if (k0 >= p.ldy) return;
  // ---- dZ'[r][k] = (sum_n dZ[r][n] W[n][k]) act'(H[r][k]) for the 16 columns k0 .., row tile = wave
  const int N = p.N;
  const bool kok = k0 + j < p.K;
  const float* __restrict__ wcol = p.W + (kok ? k0 + j : 0);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int bb = 0; bb < 2; ++bb) {
    if (16 * bb < N) {  // block-uniform
      float wf[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int n = 16 * bb + 4 * g + s;
        const bool ok = kok && n < N;
        const float wv = wcol[(long)(ok ? n : 0) * p.ldw];
        wf[s] = ok ? wv : 0.f;
      }
      const f32x4 xf = *reinterpret_cast<const f32x4*>(sDZ + (16 * wave + j) * GS_TOP_LDZ + 16 * bb + 4 * g);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = OSA_MFMA(wf[s], xf[s], acc);
    }
  }
  const int row = 16 * wave + j, k = k0 + 4 * g;
  if (row < R && k < p.ldy) {
    f32x4 h = {0.f, 0.f, 0.f, 0.f};
    if (p.act >= 0) h = *reinterpret_cast<const f32x4*>(p.aux + (long)row * p.ldaux + k);
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
  }
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
  int fold, use_max_grad_norm;
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
      m4[r] = *reinterpret_cast<const f32x4*>(M_ + ro[r]);
      v4[r] = *reinterpret_cast<const f32x4*>(V_ + ro[r]);
    }
  }
  float coef = 1.f, step_size = 0.f, ibc2 = 0.f, total_norm = 0.f, total_psq = 0.f;
  if (PHASE == 1) {
    step_size = a.fin[net * 8 + 1];
    ibc2 = a.fin[net * 8 + 2];
    if (a.fold) {
      float gq = 0.f, pq = 0.f;
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
        *reinterpret_cast<f32x4*>(M_ + ro[r]) = m;
        *reinterpret_cast<f32x4*>(V_ + ro[r]) = v;
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
