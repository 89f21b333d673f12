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
  // [3][4][H * INP]: per network a TILED private copy of W1, of its two Adam moments, and the W1 gradient of the
  // step in flight.  Tile (ft, kb) = features 16 ft .. +15 x inputs 16 kb .. +15 is 1 KB contiguous, element
  // (c, i) at c * 16 + i: a wave's 16-byte accesses to one tile are then 8 full cache lines instead of 16
  // half lines of 16 different rows (measured: the row-major version was bound by line touches, 36 k cycles
  // per step for the Adam pass alone).  Built at the start of the launch, written back at its end.
  float* ws;
};

// Phase clocks (tools/wide_pass_timing.py --phases with a -DOSA_WIDE_CLOCKS build, tools/build_variant_lib.sh):
// mean s_memtime cycles per step of thread 0 of network `net`, written over columns 0..6 of statistics row
// nmb-1-net at the end of the launch (debug builds only).
#ifdef OSA_WIDE_CLOCKS
#define WTICK(k)                                   \
  do {                                             \
    if (tid == 0) {                                \
      const long long now_ = clock64();            \
      wdbg[k] += now_ - wlast;                     \
      wlast = now_;                                \
    }                                              \
  } while (0)
#else
#define WTICK(k) do { } while (0)
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope release, which on
// gfx950 waits for EVERY outstanding vector-memory operation (one vmcnt for loads and stores): inside the slab
// loops that would drain the Adam / gradient stores just issued, and the loads that are meant to stay in flight
// across the barrier, before each barrier.  The slab loops only hand LDS tiles between the waves.
__device__ __forceinline__ void osa_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int OT>
__global__ __launch_bounds__(256, 1) void osa_wide_pass_kernel(OsaWideArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int H = 64, HT = 4, OUTP = 16 * OT;
  const OsaNet& nd = a.nd;
  const int net = blockIdx.x;
  if (!((a.nets_mask >> net) & 1)) return;
  const int KB = nd.KB, INP = nd.INP, P = nd.P;
  const int KB4 = (KB + 3) / 4;  // slabs of 4 K blocks (64 input features)
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
  const int NW1 = H * INP;
  float* __restrict__ W1 = a.ws + (long)net * 4 * NW1;  // tiled copies (see OsaWideArgs::ws)
  float* __restrict__ M1 = W1 + NW1;
  float* __restrict__ V1 = M1 + NW1;
  float* __restrict__ G1 = V1 + NW1;
  // ---- row-major parameter block -> tiled private copy (once per launch)
  for (int e = tid; e < NW1; e += 256) {
    const int t_ = e >> 8, c_ = (e >> 4) & 15, i_ = e & 15, ft_ = t_ / KB, kb_ = t_ - ft_ * KB;
    const int src = nd.oW1 + (16 * ft_ + c_) * INP + 16 * kb_ + i_;
    W1[e] = gp[src];
    M1[e] = gm[src];
    V1[e] = gv[src];
  }
  const bool critic = net != 0;
  // Observation chunks: rows are 16-byte aligned with ld % 4 == 0 (checked by the entry point), so a chunk is
  // ONE unconditional 16-byte load from an address clamped into the row, masked when it is used.  (The generic
  // osa_load_x branches per lane and the compiler then waits for every single load inside its branch: measured
  // 7 k cycles per slab of four loads in the first version of this kernel.)
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
  // the W1 elements this lane owns (D layout of the TRANSPOSED weight-gradient tiles): row 16 wave + cc,
  // columns 16 kb + 4 g .. + 3 = 16 bytes at offset cc * 16 + 4 g of tile (wave, kb) of the tiled copies
  const int w1own = wave * KB * 256 + cc * 16 + 4 * g;
  __syncthreads();  // LDS master copy + bias-correction table complete
#ifdef OSA_WIDE_CLOCKS
  long long wdbg[7] = {0, 0, 0, 0, 0, 0, 0};
  long long wlast = clock64();
#endif

#define WPUT_TILE(S, V, T)                                                           \
  _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) (S)[(16 * (T) + 4 * g + r_) * WSLD + c] = (V)[r_]

  bool pending = false;  // W1's Adam step of the previous optimiser step not applied yet
  float p_gscale = 1.f, p_step_size = 0.f, p_inv_bc2_sqrt = 1.f;
  long row_nxt;
  {
    const int B0 = (int)min((long)a.B, a.M);
    const long p0 = (16 * wave + j < B0) ? 16 * wave + j : 0;
    row_nxt = a.perm ? a.perm[p0] : p0;
  }
  for (int mb = 0; mb < a.nmb; ++mb) {
    const long mb_lo = (long)mb * a.B;
    const int Bcur = (int)(min(mb_lo + a.B, a.M) - mb_lo);
    const float invB = 1.f / (float)Bcur;
    const int c = 16 * wave + j;  // this lane's sample column
    const bool valid = c < Bcur;
    // invalid columns of a ragged minibatch gather a valid row (finite values) and meet dL/dout = 0.
    // The permutation entries of the NEXT step (this lane's sample, and the sample whose rows this thread will
    // pull into the L2) are requested now and used one step / half a step later: perm -> row -> first chunk is
    // a chain of two dependent HBM round trips that otherwise opens every step.
    const long row = row_nxt;
    const float* __restrict__ xrow = a.obs + row * ld_obs;
    long row_nn = row, prow = row;
    const bool have_next = mb + 1 < a.nmb;
    if (have_next) {
      const long nlo = mb_lo + a.B;
      const int nB = (int)(min(nlo + a.B, a.M) - nlo);
      const long np = nlo + ((c < nB) ? c : 0), pp = nlo + min(tid >> 2, nB - 1);
      row_nn = a.perm ? a.perm[np] : np;
      prow = a.perm ? a.perm[pp] : pp;
    }
    // this lane's per-sample scalars: unconditional loads issued now (clamped indices), masked where they are
    // used -- a load inside a divergent branch is waited for inside that branch
    float s_act[4 * OT], s_logp = 0.f, s_advr = 0.f, s_advc = 0.f, s_tgt = 0.f;
    if (net == 0) {  // block-uniform
      const float* __restrict__ arow = a.act + row * a.ld_act;
#pragma unroll
      for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_act[4 * o + r] = arow[min(16 * o + 4 * g + r, nd.act_dim - 1)];
      s_logp = a.logp[row];
      s_advr = a.adv_r[row];
      s_advc = a.adv_c[row];
    } else {
#pragma unroll
      for (int k = 0; k < 4 * OT; ++k) s_act[k] = 0.f;
      s_tgt = tgt[row];
    }
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
    // layer 1, fused with the W1 part of the PREVIOUS step's Adam update.  W1 is staged through LDS in slabs of
    // 4 K blocks (64 rows x 64 columns; the two X^T slab buffers of the dW1 phase are free now).  Every W1
    // element is owned by exactly one lane (tile layout, see w1own), so the lane that applies the pending Adam
    // step to its 4 x 16 bytes of a slab is also the one that writes them into the LDS slab: the forward pass
    // needs no W1 read of its own, and the memory traffic of Adam (w, m, v, g in, w, m, v out: the bandwidth-
    // bound part of the step, 21 k cycles on its own) streams one slab ahead under the 64 MFMAs of the current
    // slab.  The owners also accumulate sum w^2 of the weights this step's forward pass uses.
    float w1sq = 0.f;
    {
      f32x4 xc[4], pw[4], pm[4], pv[4], pg[4];
      auto fetch_own = [&](int s4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int off = w1own + 256 * min(4 * s4 + q, KB - 1);
          pw[q] = *reinterpret_cast<const f32x4*>(W1 + off);
          if (pending) {  // block-uniform
            pm[q] = *reinterpret_cast<const f32x4*>(M1 + off);
            pv[q] = *reinterpret_cast<const f32x4*>(V1 + off);
            pg[q] = *reinterpret_cast<const f32x4*>(G1 + off);
          }
        }
      };
      auto load_x = [&](int s4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) xc[q] = load_chunk(xrow, 16 * (4 * s4 + q) + 4 * g);
      };
      fetch_own(0);
      load_x(0);
      for (int s4 = 0; s4 < KB4; ++s4) {
        float* slab = sXs + (s4 & 1) * 64 * WSLD;
        // all arithmetic first (the 16 loads were issued a whole slab of MFMAs ago), then all stores: a store
        // issued between two waits for loads makes the next wait drain it (loads and stores share one counter)
        if (pending) {  // block-uniform
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 gg = pg[q];
            if (l2) gg = gg + pw[q] * c2;
            pw[q] = osa_adam_update4(gg * p_gscale, pm[q], pv[q], pw[q], beta1, beta2, p_step_size, p_inv_bc2_sqrt, aeps);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int kb = 4 * s4 + q;
            if (kb < KB) {  // block-uniform
              const int off = w1own + 256 * kb;
              *reinterpret_cast<f32x4*>(W1 + off) = pw[q];
              *reinterpret_cast<f32x4*>(M1 + off) = pm[q];
              *reinterpret_cast<f32x4*>(V1 + off) = pv[q];
            }
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 w = (4 * s4 + q < KB) ? pw[q] : (f32x4){0.f, 0.f, 0.f, 0.f};
          *reinterpret_cast<f32x4*>(slab + (16 * wave + cc) * WSLD + 16 * q + 4 * g) = w;
          if (critic) w1sq += (w.x * w.x + w.y * w.y) + (w.z * w.z + w.w * w.w);
        }
        f32x4 x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = mask_chunk(xc[q], 16 * (4 * s4 + q) + 4 * g);
        osa_lds_barrier();
        if (s4 + 1 < KB4) {
          fetch_own(s4 + 1);
          load_x(s4 + 1);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 w[HT];
#pragma unroll
          for (int t = 0; t < HT; ++t) w[t] = *reinterpret_cast<const f32x4*>(slab + (16 * t + i) * WSLD + 16 * q + 4 * g);
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
            for (int t = 0; t < HT; ++t) h1[t] = OSA_MFMA(w[t][s_], x[q][s_], h1[t]);
        }
      }
    }
    // touch every 128-byte line of the NEXT step's observation rows (4 threads per row; their permutation
    // entries, requested at the top of the step, have arrived by now): the gather of a permuted minibatch misses
    // the L2 (a pass streams 100 MB of rows) and an HBM round trip per slab is what the one-slab-ahead register
    // prefetch cannot hide.  The values are consumed at the end of the step.
    float pf0 = 0.f, pf1 = 0.f, pf2 = 0.f, pf3 = 0.f;
    if (have_next) {
      const float* __restrict__ pr = a.obs + prow * ld_obs;
      const int last = obs_dim - 1, pl = tid & 3;
      pf0 = pr[min(32 * pl, last)];
      pf1 = pr[min(32 * (pl + 4), last)];
      pf2 = pr[min(32 * (pl + 8), last)];
      pf3 = pr[min(32 * (pl + 12), last)];
      // ... and the lines of that sample's scalars (action row, logp, advantages / value target)
      if (net == 0) {
        const float* __restrict__ q_ = (pl == 0) ? a.act + prow * a.ld_act : (pl == 1) ? a.logp + prow
                                       : (pl == 2) ? a.adv_r + prow : a.adv_c + prow;
        pf0 += *q_;
      } else if (pl == 0) {
        pf0 += tgt[prow];
      }
    }
    WTICK(0);
    // <dW1, W1> = sum_{f,s} dz1[f][s] (a1[f][s] - b1[f]) needs the pre-activations (critics' L2 term, below)
    f32x4 pre1[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) pre1[t] = h1[t] - *reinterpret_cast<const f32x4*>(sB1 + 16 * t + 4 * g);
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
            const float z = s_act[4 * o + r] - out[o][r];
            zv[o][r] = z;
            ivar[o][r] = 1.f / var;
            lp += -(z * z) / (2.f * var) - logf(sd) - 0.91893853320467274178f;
          }
        }
      }
      lp = osa_sum_over_groups(lp);
      const float ratio = valid ? expf(lp - s_logp) : 0.f;
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
    f32x4 gw4 = {0.f, 0.f, 0.f, 0.f};
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
      gw4 = gw4 + z1[t] * pre1[t];  // <dW1, W1> = <dz1, W1 x> (critics' L2 term, see the norm below)
    }
    const float gw = (gw4.x + gw4.y) + (gw4.z + gw4.w);
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      WPUT_TILE(sDO, dO[o], o);
      WPUT_TILE(sDL, dLS[o], o);
    }
    // first X^T slab of the dW1 contraction: requested before the barrier, staged after it
    f32x4 xs[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) xs[q] = load_chunk(xrow, 16 * q + 4 * g);
    __syncthreads();  // (A) tiles complete
    WTICK(1);
    // ================= weight gradients (registers) =================
    f32x4 g2[HT], g3[OT];
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
    WTICK(2);
    // ---- dW1[f][k] = sum_s dz1[f][s] x[s][k]: X^T restaged in slabs of 4 K blocks (64 input features),
    // double-buffered; the next slab's rows are requested before the current slab's MFMAs issue
    // Every finished 16x16 tile leaves its squared norm in a register and goes to the L2-resident gradient
    // scratch (one 16-byte store per lane): holding all KB tiles in accumulators until the clip factor is
    // known needs 4 KB registers per lane and made the compiler spill for KB >= 16.
    f32x4 g1sq = {0.f, 0.f, 0.f, 0.f};
    for (int s4 = 0; s4 < KB4; ++s4) {
      float* slab = sXs + (s4 & 1) * 64 * WSLD;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 xm = mask_chunk(xs[q], 16 * (4 * s4 + q) + 4 * g);
        WPUT_TILE(slab, xm, q);
      }
      osa_lds_barrier();
      if (s4 + 1 < KB4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) xs[q] = load_chunk(xrow, 16 * (4 * (s4 + 1) + q) + 4 * g);
      }
      f32x4 acc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        f32x4 b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = *reinterpret_cast<const f32x4*>(slab + (16 * q + i) * WSLD + 16 * sb + 4 * g);
        // D[i = input 4g+r][j = feature cc] = sum_s X^T[i][s] dz1[j][s]: the lane then holds 4 CONSECUTIVE
        // inputs of one feature row of W1 -- one 16-byte access per tile and array in the Adam pass
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = OSA_MFMA(b[q][s_], a1[sb][s_], acc[q]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kb = 4 * s4 + q;
        if (kb < KB) {  // block-uniform
          g1sq = g1sq + acc[q] * acc[q];
          *reinterpret_cast<f32x4*>(G1 + w1own + 256 * kb) = acc[q];
        }
      }
    }
    WTICK(3);
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
    acc_g = acc_g + g1sq;
    float gsq = (acc_g.x + acc_g.y) + (acc_g.z + acc_g.w);
    float psq = (acc_p.x + acc_p.y) + (acc_p.z + acc_p.w);
    if (critic) {
      // W1's share of |g + 2 c w|^2 and of sum w^2 without reading W1 a second time:
      //   |g1 + c2 w1|^2 = |g1|^2 + 2 c2 <g1, w1> + c2^2 |w1|^2,   <g1, w1> = sum_{f,s} dz1[f][s] (a1[f][s] - b1[f])
      // (dW1 = dz1 x^T, so <dz1 x^T, W1> = <dz1, W1 x>); sum w1^2 was accumulated by the slab loaders
      psq += w1sq;
      if (l2) gsq += 2.f * c2 * gw + c2 * c2 * w1sq;
    }
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
    WTICK(4);
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
    // W1: clip factor and bias corrections of this step are carried into the next step's fused Adam + forward
    // loop (or the tail pass after the last step)
    pending = true;
    p_gscale = gscale;
    p_step_size = step_size;
    p_inv_bc2_sqrt = inv_bc2_sqrt;
    WTICK(5);
    // W2 / W3: weights from the LDS master copy, moments read-modify-written in global memory (L2): 24 KB per
    // step and network -- registers are what this kernel is short of
#pragma unroll
    for (int ti = 0; ti < HT; ++ti) {
      f32x4 m, v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int off = nd.oW2 + (16 * wave + 4 * g + r) * H + 16 * ti + cc;
        m[r] = gm[off];
        v[r] = gv[off];
      }
      const f32x4 w = osa_adam_update4(g2[ti] * gscale, m, v, w2r[ti], beta1, beta2, step_size, inv_bc2_sqrt, aeps);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int off = nd.oW2 + (16 * wave + 4 * g + r) * H + 16 * ti + cc;
        sW2[(16 * wave + 4 * g + r) * WSLD + 16 * ti + cc] = w[r];
        gm[off] = m[r];
        gv[off] = v[r];
      }
    }
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      f32x4 m, v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int off = nd.oW3 + (16 * o + 4 * g + r) * H + 16 * wave + cc;
        m[r] = gm[off];
        v[r] = gv[off];
      }
      const f32x4 w = osa_adam_update4(g3[o] * gscale, m, v, w3r[o], beta1, beta2, step_size, inv_bc2_sqrt, aeps);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int off = nd.oW3 + (16 * o + 4 * g + r) * H + 16 * wave + cc;
        sW3[(16 * o + 4 * g + r) * WSLD + 16 * wave + cc] = w[r];
        gm[off] = m[r];
        gv[off] = v[r];
      }
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
    asm volatile("" ::"v"(pf0), "v"(pf1), "v"(pf2), "v"(pf3));  // the prefetched lines have arrived
    row_nxt = row_nn;
    // (C) frees tiles and `red`; it also orders this step's W1 stores before the next step's W1 fragment
    // loads by the other waves: workgroup scope suffices, the four waves share one CU and its vector L1
    __syncthreads();
    WTICK(6);
  }
#undef WPUT_TILE
  // ---- tail: W1's Adam step of the last optimiser step (4 K blocks per trip, the next trip's loads in flight)
  if (pending) {
    f32x4 cw[4], cm[4], cv[4], cg[4], nw[4], nm[4], nv[4], ng[4];
    auto fetch = [&](int kb0, f32x4 (&w)[4], f32x4 (&m)[4], f32x4 (&v)[4], f32x4 (&gg)[4]) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int off = w1own + 256 * min(kb0 + q, KB - 1);
        w[q] = *reinterpret_cast<const f32x4*>(W1 + off);
        m[q] = *reinterpret_cast<const f32x4*>(M1 + off);
        v[q] = *reinterpret_cast<const f32x4*>(V1 + off);
        gg[q] = *reinterpret_cast<const f32x4*>(G1 + off);
      }
    };
    fetch(0, nw, nm, nv, ng);
    for (int kb0 = 0; kb0 < KB; kb0 += 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { cw[q] = nw[q]; cm[q] = nm[q]; cv[q] = nv[q]; cg[q] = ng[q]; }
      if (kb0 + 4 < KB) fetch(kb0 + 4, nw, nm, nv, ng);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (kb0 + q < KB) {  // block-uniform
          const int off = w1own + 256 * (kb0 + q);
          f32x4 gg = cg[q];
          if (l2) gg = gg + cw[q] * c2;
          const f32x4 w = osa_adam_update4(gg * p_gscale, cm[q], cv[q], cw[q], beta1, beta2, p_step_size,
                                           p_inv_bc2_sqrt, aeps);
          *reinterpret_cast<f32x4*>(W1 + off) = w;
          *reinterpret_cast<f32x4*>(M1 + off) = cm[q];
          *reinterpret_cast<f32x4*>(V1 + off) = cv[q];
        }
      }
    }
  }
#ifdef OSA_WIDE_CLOCKS
  __syncthreads();
  if (tid == 0 && a.nmb >= 3)  // network `net` -> row nmb-1-net, columns 0..6: mean cycles per step and phase
    for (int k = 0; k < 7; ++k) a.stats[(long)(a.nmb - 1 - net) * WNSTAT + k] = (float)wdbg[k] / (float)a.nmb;
#endif
  // ---- write back: tiled W1 / moments -> row-major parameter block, LDS master copy, bias-like Adam state
  __syncthreads();
  for (int e = tid; e < NW1; e += 256) {
    const int t_ = e >> 8, c_ = (e >> 4) & 15, i_ = e & 15, ft_ = t_ / KB, kb_ = t_ - ft_ * KB;
    const int dst = nd.oW1 + (16 * ft_ + c_) * INP + 16 * kb_ + i_;
    gp[dst] = W1[e];
    gm[dst] = M1[e];
    gv[dst] = V1[e];
  }
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

template <int OT>
static int osa_launch_wide(const OsaWideArgs& a, hipStream_t stream) {
  static OsaPerDeviceOnce attr_set;
  const size_t lds = osa_wide_lds_bytes(OT);
  if (lds > 160 * 1024) return OSA_EUNSUPPORTED;
  if (attr_set.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&osa_wide_pass_kernel<OT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return OSA_EHIP;
    attr_set.set();
  }
  hipLaunchKernelGGL((osa_wide_pass_kernel<OT>), dim3(3), dim3(256), lds, stream, a);
  return hipGetLastError() == hipSuccess ? OSA_OK : OSA_EHIP;
}

extern "C" {

int osa_ppo_wide_pass_supported(int obs_dim, int act_dim, int hidden) {
  // (from 65 inputs: the few narrow shapes osa_ppo_pass cannot hold in LDS, e.g. 90 inputs x 17 actions, come here)
  if (hidden != 64 || obs_dim < 65 || obs_dim > 512 || act_dim < 1 || act_dim > 32) return 0;
  return 1;
}

size_t osa_ppo_wide_pass_ws_floats(int obs_dim, int act_dim, int hidden) {
  if (!osa_ppo_wide_pass_supported(obs_dim, act_dim, hidden)) return 0;
  const OsaNet nd = osa_make_net(obs_dim, act_dim, hidden);
  return (size_t)3 * 4 * nd.H * nd.INP;
}

int osa_ppo_wide_pass(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                      int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                      const float* logp, const float* target_value_r, const float* target_value_c,
                      const float* adv_r, const float* adv_c, const long* perm, long M, int B,
                      const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                      float* ws, float* step_stats, void* stream) {
  if (!osa_ppo_wide_pass_supported(obs_dim, act_dim, hidden)) return OSA_EUNSUPPORTED;
  if (B > 64 || loss_kind < 0 || loss_kind > 1) return OSA_EUNSUPPORTED;  // larger batches: per-step kernels
  OSA_REQUIRE(params && adam_m && adam_v && adam_step && obs && act && logp && hp && step_stats && ws);
  OSA_REQUIRE(target_value_r && target_value_c && adv_r && adv_c && M > 0 && B > 0);
  OSA_REQUIRE(ld_obs >= obs_dim && ld_act >= act_dim);
  if (ld_obs % 4 != 0 || (reinterpret_cast<uintptr_t>(obs) & 15) != 0) return OSA_EUNSUPPORTED;  // pad the rows
  if ((double)M * ld_obs >= 2147483647.0 * 4) return OSA_EUNSUPPORTED;
  OsaWideArgs a = {};
  a.ws = ws;
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
  const int OT = a.nd.OUTP / 16;
  if (OT == 1) return osa_launch_wide<1>(a, st);
  if (OT == 2) return osa_launch_wide<2>(a, st);
  return OSA_EUNSUPPORTED;
}

}  // extern "C"
