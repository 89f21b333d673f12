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
// small minibatches (rows <= 64: the reference's YAML batch_size): the skinny path
// ------------------------------------------------------------------------------------------------
// A 64-row optimiser step at hidden 1024 is bandwidth work -- 13 MB of weights per network pass, 2 x 64 flops per weight
// byte read -- that the tiled GEMM above runs as 17 launches of 10-30 us on a few dozen workgroups each.  Here:
//   gs_fwd_kernel    Y[r][n] = act(sum_k X[r][k] W[n][k] + b[n]): one workgroup per 16 output columns and ALL rows,
//                    8 waves split the contraction; the weights are streamed ONCE, 16 bytes per lane with the whole
//                    range of a wave in flight before its first MFMA, the activations come from L2;
//   gs_bwd_kernel    dZ'[r][k] = (sum_n dZ[r][n] W[n][k]) act'(H[r][k]): one workgroup per 16 columns k, the waves
//                    split the contraction over n;
//   gs_wgrad_kernel  dW = dZ^T H of ALL layers and networks in one launch, one 64 x 64 tile per workgroup (operands via
//                    LDS, the contraction is over <= 64 rows): phase 0 adds the critics' L2 term and leaves squared-norm
//                    partials (and the gradient itself only where the caller asks for it), phase 1 REcomputes the tile
//                    -- 64 MFMAs per wave -- and applies clip + Adam straight from the accumulators: the 13 MB gradient
//                    is never written or read, a step streams weights twice and the Adam state once.
// Same arithmetic per element as the tiled path up to the summation order of the contraction (float32 MFMA chains).
struct GSProb {
  const float* X;     // fwd: input rows [R][ldx];  bwd: dZ rows [R][ldx]
  const float* W;     // [N][ldw]
  const float* bias;  // fwd: [N]
  const float* aux;   // bwd: stored outputs of the layer below [R][ldaux]
  float* Y;
  int ldx, ldw, ldy, ldaux;
  int N, K;           // W is N x K
  int act;            // fwd: activation, or -1;  bwd: activation whose derivative multiplies, or -1
};
struct GSArgs {
  GSProb p[3];
  int nprob, R;
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

// the cross-wave sum of the 4 row tiles' accumulators, in wave order; returns tile `t`'s sum for lane `ln`
__device__ __forceinline__ f32x4 gs_sum_waves(const float* red, int t, int ln) {
  f32x4 v = *reinterpret_cast<const f32x4*>(red + ((0 * 4 + t) * 64 + ln) * 4);
#pragma unroll
  for (int w = 1; w < GS_WAVES; ++w) v = v + *reinterpret_cast<const f32x4*>(red + ((w * 4 + t) * 64 + ln) * 4);
  return v;
}

// grid (ceil(maxN / 16), nprob), 64 GS_WAVES threads
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
    xrow[t] = p.X + (long)(xok[t] ? 16 * t + j : 0) * p.ldx;
  }
  for (int bb = b0; bb < b1; bb += GS_PF) {
    // every load of the window is issued before the first MFMA: the weights come from HBM, the rows from L2
    f32x4 wf[GS_PF][GS_NQ], xf[GS_PF][GS_NQ][4];
#pragma unroll
    for (int u = 0; u < GS_PF; ++u)
#pragma unroll
      for (int q = 0; q < GS_NQ; ++q)
#ifdef GS_PROBE_NOW  // (tools/skinny_probe.hip: ablations)
        wf[u][q] = (f32x4){1.f, 2.f, 3.f, 4.f};
#else
        wf[u][q] = gs_load4(wrow, GS_KB * (bb + u) + 4 * GS_NQ * g + 4 * q, p.ldw, wok && bb + u < b1);
#endif
#pragma unroll
    for (int u = 0; u < GS_PF; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < GS_NQ; ++q)
#ifdef GS_PROBE_NOX
          xf[u][q][t] = (f32x4){1.f, 2.f, 3.f, (float)t};
#elif defined(GS_PROBE_XCOAL)  // same bytes, GS_PROBE_XCOAL rows x (1024 / GS_PROBE_XCOAL) contiguous bytes per wave instruction
          xf[u][q][t] = *reinterpret_cast<const f32x4*>(
              p.X + (long)((lane / (64 / GS_PROBE_XCOAL)) + GS_PROBE_XCOAL * ((4 * (bb + u) + t) % (64 / GS_PROBE_XCOAL))) * p.ldx +
              (256 / GS_PROBE_XCOAL) * ((4 * (bb + u) + t) / (64 / GS_PROBE_XCOAL)) + 4 * (lane % (64 / GS_PROBE_XCOAL)));
#else
          xf[u][q][t] = gs_load4(xrow[t], GS_KB * (bb + u) + 4 * GS_NQ * g + 4 * q, p.ldx, xok[t] && bb + u < b1);
#endif
#pragma unroll
    for (int u = 0; u < GS_PF; ++u) {
      if (bb + u < b1) {  // wave-uniform
#pragma unroll
        for (int q = 0; q < GS_NQ; ++q)
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
#ifdef GS_PROBE_NOMFMA
            for (int t = 0; t < 4; ++t) acc[t][s] += wf[u][q][s] * xf[u][q][t][s];
#else
            for (int t = 0; t < 4; ++t) acc[t] = OSA_MFMA(wf[u][q][s], xf[u][q][t][s], acc[t]);
#endif
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
  const int k0 = blockIdx.x * 16;
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

// ---- the large layers of a skinny step: contraction split over WORKGROUPS, operands through LDS -----------------------
// gs_fwd / gs_bwd above give every workgroup 16 output columns and the WHOLE contraction: every workgroup then reads all
// 64 rows of the layer's input -- 256 KB at width 1024, 61 MB per launch over 192 workgroups, and that L2 -> L1 traffic
// (not the 12.6 MB of weights, not the MFMAs) is what a launch costs (tools/skinny_probe.hip: 14.3 us back to back, 7.7
// without the row loads).  Here a workgroup owns 64 output columns x a 256-wide SLICE of the contraction: 64 KB of weights
// + 64 KB of rows, both staged once in LDS with fully coalesced 16-byte loads and read from there as MFMA fragments (5 x
// less traffic at the same number of workgroups).  The slices' partial tiles meet in memory: every workgroup publishes
// its 64 x 64 tile (device-coherent 4-byte stores), takes a ticket, and the LAST arriver of a tile adds the slices in slice
// order -- deterministic, whatever the arrival order -- and applies the epilogue.  Tickets return to zero.
#define GSB_CL 256             // contraction columns per slice
#define GSB_LD (GSB_CL + 4)    // leading dimension of the [64][256] operand images in LDS
#define GSB_LDT 68             // leading dimension of the [256][64] image (backward: W rows are contiguous along the output)
struct GSBArgs {
  GSProb p[3];
  int nprob, R;
  int tiles[3], S[3];          // 64-column output tiles and contraction slices of every problem
  float* slab;                 // [sum tiles x S][64 rows][64 columns] partial tiles
  int* ticket;                 // [sum tiles], zero between launches
};

template <bool BWD>
__global__ __launch_bounds__(512) void gs_big_kernel(GSBArgs a) {
  extern __shared__ __attribute__((aligned(16))) float gsb_smem[];
  float* sA = gsb_smem;                                  // FWD: W slice [64 cols][GSB_LD]; BWD: W slice [256 n][GSB_LDT]
  float* sB = gsb_smem + (BWD ? GSB_CL * GSB_LDT : 64 * GSB_LD);  // rows [64][GSB_LD]
  __shared__ int s_last;
  // ---- which (problem, tile, slice)
  int b = blockIdx.x, pi = 0, tbase = 0;
#pragma unroll
  for (int q = 0; q < 2; ++q)
    if (pi == q && q + 1 < a.nprob && b >= a.tiles[q] * a.S[q]) {
      b -= a.tiles[q] * a.S[q];
      tbase += a.tiles[q] * a.S[q];
      pi = q + 1;
    }
  const GSProb p = pi == 0 ? a.p[0] : (pi == 1 ? a.p[1] : a.p[2]);
  const int S = pi == 0 ? a.S[0] : (pi == 1 ? a.S[1] : a.S[2]);
  const int tile = b / S, sl = b - tile * S;
  const int tkidx = tile + (pi == 0 ? 0 : (pi == 1 ? a.tiles[0] : a.tiles[0] + a.tiles[1]));  // the tile's ticket
  const int c0 = 64 * tile;      // first output column
  const int q0 = GSB_CL * sl;    // first contraction index
  const int C = BWD ? p.N : p.K; // contraction length;  outputs: FWD p.N columns, BWD p.K columns
  const int NO = BWD ? p.K : p.N;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int R = a.R;
  // ---- stage the operands (16-byte loads, a wave on whole rows)
  if (!BWD) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {  // 64 rows x 64 pieces
      const int u = tid + 512 * q, r = u >> 6, c4 = 4 * (u & 63);
      const bool okw = c0 + r < p.N;
      *reinterpret_cast<f32x4*>(sA + r * GSB_LD + c4) =
          gs_load4(p.W + (long)(okw ? c0 + r : 0) * p.ldw, q0 + c4, p.ldw, okw);
      *reinterpret_cast<f32x4*>(sB + r * GSB_LD + c4) = gs_load4(p.X + (long)(r < R ? r : 0) * p.ldx, q0 + c4, p.ldx, r < R);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) {  // W: 256 rows n x 16 pieces;  dZ: 64 rows x 64 pieces
      const int u = tid + 512 * q;
      const int n = u >> 4, c4 = 4 * (u & 15);
      const bool okw = q0 + n < p.N;
      *reinterpret_cast<f32x4*>(sA + n * GSB_LDT + c4) =
          gs_load4(p.W + (long)(okw ? q0 + n : 0) * p.ldw, c0 + c4, p.ldw, okw);
      const int r = u >> 6, d4 = 4 * (u & 63);
      *reinterpret_cast<f32x4*>(sB + r * GSB_LD + d4) = gs_load4(p.X + (long)(r < R ? r : 0) * p.ldx, q0 + d4, p.ldx, r < R);
    }
  }
  __syncthreads();
  // ---- MFMA: wave = (column tile cw of 16, half hw of the slice); D[m = column][n = row] as in gs_fwd / gs_bwd
  const int cw = wave & 3, hw = wave >> 2;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int bb = 0; bb < GSB_CL / 32; ++bb) {
    const int k = (GSB_CL / 2) * hw + 16 * bb + 4 * g;
    f32x4 af, bf[4];
    if (!BWD) {
      af = *reinterpret_cast<const f32x4*>(sA + (16 * cw + j) * GSB_LD + k);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) af[s] = sA[(k + s) * GSB_LDT + 16 * cw + j];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) bf[t] = *reinterpret_cast<const f32x4*>(sB + (16 * t + j) * GSB_LD + k);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = OSA_MFMA(af[s], bf[t][s], acc[t]);
  }
  __syncthreads();  // operand images consumed: sB becomes the [row][64] output tile
  // ---- the two halves of the slice, added in LDS; tile element (row, column) at sB[row * GSB_LDT + column]
  float* sT = sB;
  if (hw == 1) {
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(sT + (16 * t + j) * GSB_LDT + 16 * cw + 4 * g) = acc[t];
  }
  __syncthreads();
  if (hw == 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4* d = reinterpret_cast<f32x4*>(sT + (16 * t + j) * GSB_LDT + 16 * cw + 4 * g);
      *d = acc[t] + *d;
    }
  }
  __syncthreads();
  // ---- publish / combine.  Thread -> (row, 4 columns): 64 rows x 16 pieces = 1024 pieces, two per thread
  float* __restrict__ myslab = a.slab + ((long)(tbase + tile * S + sl)) * 4096;
  if (S > 1) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int u = tid + 512 * q, r = u >> 4, c4 = 4 * (u & 15);
      const f32x4 v = *reinterpret_cast<const f32x4*>(sT + r * GSB_LDT + c4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        __hip_atomic_store(myslab + r * 64 + c4 + e, v[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the tile's stores have been performed at the device-coherent level before the ticket is taken
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) {
      const int seen = __hip_atomic_fetch_add(a.ticket + tkidx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = seen == S - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int u = tid + 512 * q, r = u >> 4, c4 = 4 * (u & 15);
    f32x4 v;
    if (S > 1) {
      const float* __restrict__ s0 = a.slab + ((long)(tbase + tile * S)) * 4096 + r * 64 + c4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = __hip_atomic_load(s0 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int k = 1; k < S; ++k)
          t += __hip_atomic_load(s0 + (long)k * 4096 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v[e] = t;
      }
    } else {
      v = *reinterpret_cast<const f32x4*>(sT + r * GSB_LDT + c4);
    }
    const int col = c0 + c4;
    if (r < R && col < p.ldy) {
      f32x4 h = {0.f, 0.f, 0.f, 0.f};
      if (BWD && p.act >= 0) h = *reinterpret_cast<const f32x4*>(p.aux + (long)r * p.ldaux + col);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y = 0.f;  // (padding columns of the row: zero)
        if (col + e < NO) {
          y = v[e];
          if (!BWD) {
            if (p.bias) y += p.bias[col + e];
            if (p.act >= 0) y = gm_act(y, p.act);
          } else if (p.act >= 0) {
            y *= gm_dact(h[e], p.act);
          }
        }
        v[e] = y;
      }
      *reinterpret_cast<f32x4*>(p.Y + (long)r * p.ldy + col) = v;
    }
  }
  (void)C;
  if (S > 1 && tid == 0) {
    __hip_atomic_store(a.ticket + tkidx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
  }
}

struct GSWLayer {
  const float* dZ;  // [R][ldz]   dL/d(pre-activation) of this layer
  const float* H;   // [R][ldh]   the layer's input rows
  int ldz, ldh, out, in, ldw, oW, ob, tile0, tk;
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
    *reinterpret_cast<f32x4*>(sH + r * GS_LDT + c4) = gs_load4(ly.H + (long)(r < R ? r : 0) * ly.ldh, k0 + c4, ly.ldh, r < R);
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
