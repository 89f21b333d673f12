// General actor-critic MLPs on gfx950 matrix cores: ANY `hidden_sizes` the reference's model builder accepts
// (omnisafe/utils/model.py:73-111: any depth, any widths, actor and critics independent; the reference's one
// published timing table has a 1024 x 1024 row, docs/source/start/efficiency.rst:15-23), where the fused kernels of
// mlp_kernels.hip / ppo_pass_body.h cover [H, H] with H <= 256.
//
// Networks this wide do not fit a compute unit: a 1024 x 1024 layer is 4 MB of weights, so every layer is a real
// GEMM over the minibatch and the path is layer-wise:
//   forward     H_l  = act(H_{l-1} W_l^T + b_l)                          C[M = rows][N = out]  "NT"
//   backward    dZ_{l-1} = (dZ_l W_l) * act'(H_{l-1})                     C[rows][in]           "NN"
//               dW_l = dZ_l^T H_{l-1},  db_l = dZ_l^T 1                   C[out][in (+ 1)]      "TN", split over rows
// all three on ONE hand-written float32 MFMA kernel (v_mfma_f32_32x32x2_f32: exact float32, bit-for-bit an fmaf
// chain, 157 TFLOP/s peak): 128 x 128 (or 64-wide) output tile per 256-thread workgroup, K step 16 staged through
// LDS ([row][k] images with a 20-float leading dimension: the 16-byte fragment reads of a lane group hit 16 distinct
// bank quads), global -> register -> LDS double buffering, four waves as 2 x 2 with 2 x 2 MFMA tiles each, up to three
// problems (the three networks) per launch, fused epilogues (bias + activation; act' of the stored output; C +=;
// the bias gradient as one more output column fed by a column of ones).  Loss, clip and Adam are small elementwise /
// reduction kernels around it; every reduction has a fixed order (slabs, block partials): deterministic.
//
// What bounds it: at hidden 1024 and batch 16 384 a step is 3 networks x 6 W rows = 2 x 10^11 FLOP: MFMA-bound
// (roofline in bench.py --hidden 1024); at the YAML batch of 64 rows it streams 3 x 13 MB of weights, moments and
// gradients per step: HBM / L2-bound.
#include <stdlib.h>

#include "skinny_mlp.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GM_NSTAT 16

namespace {

struct GNet {
  int L;                                   // linear layers (hidden layers + output layer)
  int in[GM_MAXL], out[GM_MAXL];           // layer l: out[l] x in[l]
  int ld[GM_MAXL];                         // leading dimension of W_l rows (in[l] rounded up to 4)
  int ldh[GM_MAXL];                        // leading dimension of the layer's output rows (out[l] rounded up to 4)
  int oW[GM_MAXL], ob[GM_MAXL];            // offsets into the network's parameter block
  int oLS;                                 // actor: log_std; critics: end of the real parameters
  int Pn;                                  // floats used by this network (<= P)
  int act;                                 // hidden activation (OSA_ACT_*)
  int maxldh;                              // widest layer output
};
struct GLayout {
  GNet n[3];
  int P, obs_dim, act_dim, ldx, lda;       // block stride; gathered observation / action row strides
};

inline int r4(int x) { return (x + 3) / 4 * 4; }

int gm_make_layout(const osa_gmlp_desc* d, GLayout* lo) {
  if (!d || d->obs_dim < 1 || d->act_dim < 1) return OSA_EINVAL;
  lo->obs_dim = d->obs_dim;
  lo->act_dim = d->act_dim;
  lo->ldx = r4(d->obs_dim);
  lo->lda = r4(d->act_dim);
  int pmax = 0;
  for (int net = 0; net < 3; ++net) {
    GNet& g = lo->n[net];
    g.L = d->n_layers[net];
    if (g.L < 1 || g.L > GM_MAXL) return OSA_EUNSUPPORTED;
    if (d->activation[net] < OSA_ACT_TANH || d->activation[net] > OSA_ACT_IDENTITY) return OSA_EUNSUPPORTED;
    g.act = d->activation[net];
    int off = 0, prev = d->obs_dim;
    g.maxldh = 4;
    for (int l = 0; l < g.L; ++l) {
      const int w = d->width[net][l];
      if (w < 1) return OSA_EINVAL;
      g.in[l] = prev;
      g.out[l] = w;
      g.ld[l] = r4(prev);
      g.ldh[l] = r4(w);
      if (g.ldh[l] > g.maxldh) g.maxldh = g.ldh[l];
      g.oW[l] = off;
      off += w * g.ld[l];
      g.ob[l] = off;
      off += r4(w);
      prev = w;
    }
    if (prev != (net == 0 ? d->act_dim : 1)) return OSA_EINVAL;  // the last width is the output width
    g.oLS = off;
    if (net == 0) off += r4(d->act_dim);
    g.Pn = off;
    if (off > pmax) pmax = off;
  }
  lo->P = (pmax + 15) / 16 * 16;
  return OSA_OK;
}

// ---- workspace carve-up (floats; every offset a multiple of 4 -> 16-byte aligned rows)
#define GM_ROW_SPLIT_ROWS 256  // minibatches up to this many rows split the K of their forward / backward-data GEMMs
#define GM_ROW_SPLITS 8
struct GWs {
  size_t xg, actg, scal;          // gathered observations [R][ldx], actions [R][lda], scalars [6][R]
  size_t h[3][GM_MAXL];           // layer outputs of every network [R][ldh]
  size_t t[GM_MAXL];              // tangent layer outputs of the actor (Fisher-vector product)
  size_t z[3][2];                 // ping-pong dL/d(pre-activation) [R][maxldh]
  size_t dls, lpart;              // per-block partial sums: log_std gradient [nblk][lda], loss statistics [3][nblk][4]
  size_t slab[3][GM_MAXL];        // partial gradients of layer l: [S_l][W block + bias block]
  int S[3][GM_MAXL];              // splits of the row (= reduction) dimension of layer l's weight-gradient GEMM
  size_t npart, fin;              // norm partials [3][nb][2]; finals [3][8]
  size_t dws;                     // double scratch of the KL / evaluation reductions (4 x 1024 doubles)
  size_t rsl, rsl_floats;         // K-slice slabs of the small-row forward / backward-data GEMMs (gm_gemm_rows)
  // skinny path (R <= GS_MAX_ROWS): dL/d(pre-activation) of EVERY layer stays alive until the weight-gradient launch;
  // its 64 x 64 tiles, network by network and layer by layer (+ one tail workgroup per network)
  size_t zs[3][GM_MAXL];
  int sk_tile0[3][GM_MAXL], sk_tk[3][GM_MAXL], sk_ntile[3], sk_maxT1;
  // (round 6) the clip norm from the forward / backward launches (skinny_mlp.h: gs_gram_norm): outputs before the bias,
  // Gram matrices of the layers' input rows, partial outputs of the fused top layer, norm slots [3][slot_stride][4]
  size_t ylin[3][GM_MAXL], gram[3][GM_MAXL], oslab[3], slots;
  int fs[3][GM_MAXL], ds[3][GM_MAXL], ts[3], nslot[3], slot_stride;
  size_t total;
  int nblk, nb;
};
#define GS_MAX_ROWS 64

// Splits of the row (= reduction) dimension of ONE layer's weight-gradient GEMM: the launch (three networks) should
// fill the chip -- output tiles x networks x splits >= ~512 workgroups -- with at least 128 rows per split and at most
// 32 splits.  Per layer: the skinny first / last layers (1024 x 61, 2 x 1025) have a handful of output tiles and
// need many splits, the square ones few (a global split count chosen for the largest layer left them on 48
// workgroups: 1.35 of the 5.3 ms of a 1024 x 1024 step at 16 384 rows).
int gm_splits(const GNet& n, int l, long R) {
  const int tm = n.out[l] > 64 ? 128 : 64, tn = n.in[l] + 1 > 64 ? 128 : 64;
  const long tiles = (long)((n.out[l] + tm - 1) / tm) * ((n.in[l] + 1 + tn - 1) / tn);
  long s = (512 + tiles * 3 - 1) / (tiles * 3);
  const long smax = R / 128;
  if (s > smax) s = smax;
  if (s > 32) s = 32;
  if (s < 1) s = 1;
  return (int)s;
}

GWs gm_ws(const GLayout& lo, long R) {
  GWs w;
  size_t off = 0;
  auto take = [&](size_t n) {
    const size_t o = off;
    off += (n + 3) / 4 * 4;
    return o;
  };
  w.xg = take((size_t)R * lo.ldx);
  w.actg = take((size_t)R * lo.lda);
  w.scal = take((size_t)6 * R);
  for (int net = 0; net < 3; ++net)
    for (int l = 0; l < GM_MAXL; ++l) w.h[net][l] = l < lo.n[net].L ? take((size_t)R * lo.n[net].ldh[l]) : 0;
  for (int l = 0; l < GM_MAXL; ++l) w.t[l] = l < lo.n[0].L ? take((size_t)R * lo.n[0].ldh[l]) : 0;
  for (int net = 0; net < 3; ++net)
    for (int k = 0; k < 2; ++k) w.z[net][k] = take((size_t)R * lo.n[net].maxldh);
  w.nblk = (int)((R + 255) / 256);
  w.dls = take((size_t)w.nblk * lo.lda);
  w.lpart = take((size_t)3 * w.nblk * 4);
  for (int net = 0; net < 3; ++net)
    for (int l = 0; l < GM_MAXL; ++l) {
      w.S[net][l] = 0;
      w.slab[net][l] = 0;
      if (l >= lo.n[net].L) continue;
      const GNet& n = lo.n[net];
      w.S[net][l] = gm_splits(n, l, R);
      w.slab[net][l] = take((size_t)w.S[net][l] * (n.out[l] * n.ld[l] + r4(n.out[l])));
    }
  w.nb = (lo.P + 1023) / 1024;
  w.sk_maxT1 = 0;
  for (int net = 0; net < 3; ++net) {
    int t = 0;
    for (int l = 0; l < GM_MAXL; ++l) {
      w.zs[net][l] = 0;
      w.sk_tile0[net][l] = t;
      w.sk_tk[net][l] = 1;
      if (l >= lo.n[net].L || R > GS_MAX_ROWS) continue;
      const GNet& n = lo.n[net];
      w.zs[net][l] = take((size_t)R * n.ldh[l]);
      w.sk_tk[net][l] = (n.ld[l] + 63) / 64;
      t += ((n.out[l] + 63) / 64) * w.sk_tk[net][l];
    }
    w.sk_ntile[net] = t;
    if (t + 1 > w.sk_maxT1) w.sk_maxT1 = t + 1;
  }
  w.slot_stride = 0;
  for (int net = 0; net < 3; ++net) {
    const GNet& n = lo.n[net];
    int c = 0;
    w.oslab[net] = 0;
    w.ts[net] = 0;
    for (int l = 0; l < GM_MAXL; ++l) {
      w.ylin[net][l] = w.gram[net][l] = 0;
      w.fs[net][l] = w.ds[net][l] = 0;
      if (l + 1 >= n.L || R > GS_MAX_ROWS) continue;  // (layers below the top layer)
      w.ylin[net][l] = take((size_t)R * n.ldh[l]);
      w.gram[net][l] = take(64 * 64);
      const int tiles = (n.ldh[l] + 15) / 16;
      w.fs[net][l] = c; c += tiles;
      w.ds[net][l] = c; c += tiles;
    }
    if (n.L >= 2 && R <= GS_MAX_ROWS) {
      const int tiles = (n.ldh[n.L - 2] + 15) / 16;
      w.ts[net] = c; c += tiles;
      w.oslab[net] = take((size_t)tiles * 64 * n.ldh[n.L - 1]);
    }
    w.nslot[net] = c;
    if (c > w.slot_stride) w.slot_stride = c;
  }
  w.slots = take((size_t)3 * w.slot_stride * 4);
  w.npart = take((size_t)3 * (w.nb > w.sk_maxT1 ? w.nb : w.sk_maxT1) * 2);
  w.fin = take(3 * 8);
  w.dws = take(2 * 4 * 1024);
  {  // small minibatches: [3 networks][GM_ROW_SPLITS][R][widest layer]
    int maxld = lo.ldx;
    for (int net = 0; net < 3; ++net)
      for (int l = 0; l < lo.n[net].L; ++l)
        if (lo.n[net].ldh[l] > maxld) maxld = lo.n[net].ldh[l];
    w.rsl_floats = R <= GM_ROW_SPLIT_ROWS ? (size_t)3 * GM_ROW_SPLITS * R * maxld : 0;
    w.rsl = take(w.rsl_floats);
  }
  w.total = off;
  return w;
}

// ------------------------------------------------------------------------------------------------
// the GEMM
// ------------------------------------------------------------------------------------------------
struct GProb {
  const float* A;
  const float* B;
  float* C;
  const float* bias;   // epilogue: + bias[n]
  const float* aux;    // epilogue: * act'(aux[m][n])
  float* cb;           // ones_n >= 0: column ones_n of the product goes to cb[m] (the bias gradient)
  int M, N, K, lda, ldb, ldc, ldaux;
  int act;             // epilogue: activation code, or -1
  int dact;            // epilogue: activation code whose derivative multiplies, or -1
  int accumulate;      // C += product
  int ones_n;          // >= 0: B(k, ones_n) = 1 for every k (only with BT)
  int cb_pad;          // cb[M .. cb_pad) = 0 (padding of the bias block)
  long slab_stride;    // split s writes C + s * slab_stride and cb + s * slab_stride
  int splits;          // this problem's splits of K (<= GArgs.splits, the launch's grid.z per problem)
};
struct GArgs {
  GProb p[3];
  int nprob, splits;
  // != 0: XCD-aware tile order (+ 1 % at 1024 x 1024 / 16 384 rows: 3 960 v 3 998 us per step, same box).  Workgroup i of a launch lands on XCC i mod 8 (each with its own L2): in dispatch order
  // (x fastest) the 8 column tiles of a 1024-wide layer go to 8 DIFFERENT L2s and every one of them streams the whole
  // row operand.  Remapped, XCC c owns the row blocks c, c + 8, ... with ALL their column tiles: a row block is fetched
  // into one L2 once, the (small) column operand into every L2.  Same tiles, same arithmetic.
  int xcd;
};

// C[m][n] = epi(sum_k A(m, k) B(k, n)).   AT: A(m, k) = A[k lda + m] (else A[m lda + k]);
//                                          BT: B(k, n) = B[k ldb + n] (else B[n ldb + k]).
// BK: K step per LDS stage.  16 when the launch has enough workgroups to hide memory latency behind each other; 64
// when it has few (the skinny GEMMs of a 64-row minibatch stream megabytes of weights through a few dozen workgroups:
// a step of 16 with one tile of lookahead paid a full L2 round trip per 0.2 us of MFMA work -- 40-70 us per layer at
// hidden 1024; four times the bytes in flight per barrier).
template <int TM, int TN, int BK, bool AT, bool BT>
__global__ __launch_bounds__(256) void gm_gemm_kernel(GArgs g) {
  constexpr int LD = BK + 4, MI = TM / 64, NJ = TN / 64, UA = TM * BK / 1024, UB = TN * BK / 1024;
  // An operand that is contiguous along its tile-row index (AT / BT: the activations of the weight-gradient GEMM, the
  // weights of the backward-data GEMM) stays K-MAJOR in LDS -- [k][row], leading dimension rows + 8 -- instead of being
  // transposed on the way in: 16-byte global loads with a wave on 1 KB of consecutive bytes and 16-byte LDS stores, where
  // the transposing stores were four scalar writes per piece from 64-byte row segments; the MFMA fragments then are four
  // scalar LDS reads of 32 consecutive floats per lane group (the two lane groups 4 rows = 32 banks apart) instead of
  // one 16-byte read.  Same values into the same MFMAs: same bits.
  constexpr int LDTA = TM + 8, LDTB = TN + 8;
  constexpr int SZA = AT ? BK * LDTA : TM * LD, SZB = BT ? BK * LDTB : TN * LD;
  extern __shared__ __attribute__((aligned(16))) float gm_smem[];
  float(*sA)[SZA] = reinterpret_cast<float(*)[SZA]>(gm_smem);
  float(*sB)[SZB] = reinterpret_cast<float(*)[SZB]>(gm_smem + 2 * SZA);
  const int pi = blockIdx.z / g.splits, sp = blockIdx.z - pi * g.splits;
  // (selected with scalar moves: a run-time index into the by-value argument would go through scratch memory)
  const GProb p = pi == 0 ? g.p[0] : (pi == 1 ? g.p[1] : g.p[2]);
  const int M = p.M, N = p.N, K = p.K;
  const int ncols = N + (p.ones_n >= 0 ? 1 : 0);
  int bx = blockIdx.x, by = blockIdx.y;
  if (g.xcd && (gridDim.y & 7) == 0) {
    const int gx = gridDim.x, lid = bx + gx * by, xcd = lid & 7, slot = lid >> 3;
    by = (slot / gx) * 8 + xcd;
    bx = slot - (slot / gx) * gx;
  }
  const int m0 = by * TM, n0 = bx * TN;
  if (m0 >= M || n0 >= (ncols > p.ldc ? ncols : p.ldc) || sp >= p.splits) return;
  int kper = (K + p.splits - 1) / p.splits;
  kper = (kper + BK - 1) / BK * BK;
  const int kbeg = sp * kper;
  const int kend = K < kbeg + kper ? K : kbeg + kper;
  const int nkt = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kk = lane >> 5;
  const int wm = (wave >> 1) * (TM / 2), wn = (wave & 1) * (TN / 2);
  const float* __restrict__ A = p.A;
  const float* __restrict__ B = p.B;
  f32x4 ra[UA], rb[UB];
  auto load_tile = [&](int kt) {
    const int k0 = kbeg + kt * BK;
#pragma unroll
    for (int q = 0; q < UA; ++q) {
      const int u = tid + 256 * q;
      if constexpr (!AT) {
        const int r = u / (BK / 4), c = u % (BK / 4), m = m0 + r;
        ra[q] = gm_load4(A + (long)m * p.lda, k0 + 4 * c, kend, m < M);
      } else {
        const int kq = u / (TM / 4), c = u % (TM / 4), k = k0 + kq;
        ra[q] = gm_load4(A + (long)k * p.lda, m0 + 4 * c, M, k < kend);
      }
    }
#pragma unroll
    for (int q = 0; q < UB; ++q) {
      const int u = tid + 256 * q;
      if constexpr (!BT) {
        const int r = u / (BK / 4), c = u % (BK / 4), n = n0 + r;
        rb[q] = gm_load4(B + (long)n * p.ldb, k0 + 4 * c, kend, n < N);
      } else {
        const int kq = u / (TN / 4), c = u % (TN / 4);
        const int k = k0 + kq, nn = n0 + 4 * c;
        f32x4 v = gm_load4(B + (long)k * p.ldb, nn, N, k < kend);
        if (p.ones_n >= 0 && k < kend) {  // the column of ones that turns the bias gradient into one more output column
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (nn + i == p.ones_n) v[i] = 1.f;
        }
        rb[q] = v;
      }
    }
  };
  auto store_tile = [&](int st) {
#pragma unroll
    for (int q = 0; q < UA; ++q) {
      const int u = tid + 256 * q;
      if constexpr (!AT) {
        *reinterpret_cast<f32x4*>(&sA[st][(u / (BK / 4)) * LD + 4 * (u % (BK / 4))]) = ra[q];
      } else {
        *reinterpret_cast<f32x4*>(&sA[st][(u / (TM / 4)) * LDTA + 4 * (u % (TM / 4))]) = ra[q];
      }
    }
#pragma unroll
    for (int q = 0; q < UB; ++q) {
      const int u = tid + 256 * q;
      if constexpr (!BT) {
        *reinterpret_cast<f32x4*>(&sB[st][(u / (BK / 4)) * LD + 4 * (u % (BK / 4))]) = rb[q];
      } else {
        *reinterpret_cast<f32x4*>(&sB[st][(u / (TN / 4)) * LDTB + 4 * (u % (TN / 4))]) = rb[q];
      }
    }
  };
  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (nkt > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int st = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);  // the next tile's global loads are in flight under this tile's MFMAs
#pragma unroll
    for (int s = 0; s < BK / 8; ++s) {
      // K permutation inside a block of 8: lane group kk consumes k = 8 s + 4 kk + j in MFMA step j, for A and B
      // alike -- each fragment is ONE 16-byte LDS read per lane and feeds four MFMAs
      f32x4 af[MI], bf[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        if constexpr (!AT) {
          af[i] = *reinterpret_cast<const f32x4*>(&sA[st][(wm + 32 * i + l31) * LD + 8 * s + 4 * kk]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) af[i][q] = sA[st][(8 * s + 4 * kk + q) * LDTA + wm + 32 * i + l31];
        }
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if constexpr (!BT) {
          bf[j] = *reinterpret_cast<const f32x4*>(&sB[st][(wn + 32 * j + l31) * LD + 8 * s + 4 * kk]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) bf[j][q] = sB[st][(8 * s + 4 * kk + q) * LDTB + wn + 32 * j + l31];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q], bf[j][q], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) store_tile(st ^ 1);
    __syncthreads();
  }
  // ---- epilogue: D[row = (r & 3) + 8 (r >> 2) + 4 kk][col = l31] of every 32 x 32 tile
  float* __restrict__ C = p.C + (long)sp * p.slab_stride;
  float* __restrict__ cb = p.cb ? p.cb + (long)sp * p.slab_stride : nullptr;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n0 + wn + 32 * j + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kk;
        float v = acc[i][j][r];
        if (m < M) {
          if (n < N) {
            float* dst = C + (long)m * p.ldc + n;
            if (p.accumulate) v += *dst;
            if (p.bias) v += p.bias[n];
            if (p.act >= 0) v = gm_act(v, p.act);
            if (p.dact >= 0) v *= gm_dact(p.aux[(long)m * p.ldaux + n], p.dact);
            *dst = v;
          } else {
            if (n == p.ones_n && cb) cb[m] = v;
            if (n < p.ldc && p.ones_n >= 0) C[(long)m * p.ldc + n] = 0.f;  // padding columns of the weight block
          }
        } else if (n == p.ones_n && cb && m < p.cb_pad) {
          cb[m] = 0.f;  // padding of the bias block
        }
      }
    }
}

template <int TM, int TN, int BK, bool AT, bool BT>
int gm_launch(const GArgs& g_, int maxM, int maxN, hipStream_t st) {
  static const int xcd_sw = [] {  // (A/B switch: OSA_GMLP_XCD_ORDER=0 keeps the dispatch order)
    const char* v = getenv("OSA_GMLP_XCD_ORDER");
    return (v != nullptr && v[0] == '0' && v[1] == 0) ? 0 : 1;
  }();
  GArgs g = g_;
  g.xcd = xcd_sw;
  constexpr size_t lds = (size_t)2 * ((AT ? BK * (TM + 8) : TM * (BK + 4)) + (BT ? BK * (TN + 8) : TN * (BK + 4))) * sizeof(float);
  if constexpr (lds > 64 * 1024) {
    static OsaPerDeviceOnce attr_set;
    if (attr_set.need()) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gm_gemm_kernel<TM, TN, BK, AT, BT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return OSA_EHIP;
      attr_set.set();
    }
  }
  dim3 grid((maxN + TN - 1) / TN, (maxM + TM - 1) / TM, g.nprob * g.splits);
  hipLaunchKernelGGL((gm_gemm_kernel<TM, TN, BK, AT, BT>), grid, dim3(256), lds, st, g);
  return hipGetLastError() == hipSuccess ? OSA_OK : OSA_EHIP;
}

#ifndef GM_BK
#define GM_BK 16  // K step of the launches with enough workgroups (A/B: -DGM_BK=32)
#endif
// one grouped launch; the tile follows the largest problem (64-wide tiles for skinny ones), the K step the number of
// workgroups the launch will have
template <bool AT, bool BT>
int gm_gemm(const GArgs& g, hipStream_t st) {
  int maxM = 0, maxN = 0;
  for (int i = 0; i < g.nprob; ++i) {
    const int nc = g.p[i].ones_n >= 0 ? (g.p[i].ldc > g.p[i].N + 1 ? g.p[i].ldc : g.p[i].N + 1) : g.p[i].N;
    if (g.p[i].M > maxM) maxM = g.p[i].M;
    if (nc > maxN) maxN = nc;
  }
  if (g.nprob == 0 || maxM == 0) return OSA_OK;
  bool bigM = maxM > 64, bigN = maxN > 64;
  auto count = [&](int tm, int tn) {
    return (long)((maxM + tm - 1) / tm) * ((maxN + tn - 1) / tn) * g.nprob * g.splits;
  };
  // a launch that would leave most of the chip idle takes the 64-wide tiles (twice / four times the workgroups)
  if (bigN && count(bigM ? 128 : 64, 128) < 128) bigN = false;
  if (bigM && count(128, bigN ? 128 : 64) < 128) bigM = false;
  const long wgs = count(bigM ? 128 : 64, bigN ? 128 : 64);
  static const int deep_sw = [] {  // (A/B switch: OSA_GMLP_DEEP=0 / 1 forces the K step 16 / 64)
    const char* v = getenv("OSA_GMLP_DEEP");
    return v == nullptr ? -1 : (v[0] == '1' ? 1 : 0);
  }();
  const bool deep = deep_sw >= 0 ? deep_sw == 1 : wgs < 256;  // few workgroups: latency-bound -> K step 64
#define GM_GO(TM_, TN_)                                                              \
  return deep ? gm_launch<TM_, TN_, 64, AT, BT>(g, maxM, maxN, st) : gm_launch<TM_, TN_, GM_BK, AT, BT>(g, maxM, maxN, st)
  if (bigM && bigN) GM_GO(128, 128);
  if (bigM) GM_GO(128, 64);
  if (bigN) GM_GO(64, 128);
  GM_GO(64, 64);
#undef GM_GO
}

// ---- small minibatches: the forward / backward-data GEMMs of a 64-row step are [64 x K] x [K x N] products whose K
// loop streams megabytes of weights through a few dozen workgroups -- 16 dependent K steps with one tile of lookahead
// and 48 workgroups' worth of bytes in flight: 25 us per layer at hidden 1024, whatever the arithmetic.  Here the K
// range is cut into up to 8 slices (one workgroup each: 8 x the bytes in flight), the slices' raw products go to slabs
// and ONE small launch adds them in slice order and applies the epilogue (bias, activation, activation' x aux).
struct GEpi {
  const float* slab;
  float* C;
  const float* bias;
  const float* aux;
  int M, N, ldc, ldaux, act, dact, S;
  long slab_stride;
};
struct GEpiArgs {
  GEpi e[3];
  int n;
};
__global__ __launch_bounds__(256) void gm_rows_epilogue_kernel(GEpiArgs a) {
  const GEpi e = blockIdx.y == 0 ? a.e[0] : (blockIdx.y == 1 ? a.e[1] : a.e[2]);
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)e.M * e.N) return;
  const int m = (int)(idx / e.N), n = (int)(idx - (long)m * e.N);
  const float* s = e.slab + (long)m * e.ldc + n;
  float t[GM_ROW_SPLITS];
#pragma unroll
  for (int k = 0; k < GM_ROW_SPLITS; ++k) t[k] = k < e.S ? s[(long)k * e.slab_stride] : 0.f;
  float v = t[0];
#pragma unroll
  for (int k = 1; k < GM_ROW_SPLITS; ++k)
    if (k < e.S) v += t[k];
  if (e.bias) v += e.bias[n];
  if (e.act >= 0) v = gm_act(v, e.act);
  if (e.dact >= 0) v *= gm_dact(e.aux[(long)m * e.ldaux + n], e.dact);
  e.C[(long)m * e.ldc + n] = v;
}

template <bool AT, bool BT>
int gm_gemm_rows(GArgs g, float* scratch, size_t scratch_floats, hipStream_t st) {
  int maxM = 0, maxN = 0, maxK = 0;
  for (int i = 0; i < g.nprob; ++i) {
    if (g.p[i].M > maxM) maxM = g.p[i].M;
    if (g.p[i].N > maxN) maxN = g.p[i].N;
    if (g.p[i].K > maxK) maxK = g.p[i].K;
  }
  static const bool off = [] {
    const char* v = getenv("OSA_GMLP_ROW_SPLIT");
    return v != nullptr && v[0] == '0' && v[1] == 0;
  }();
  if (off || g.nprob == 0 || maxM > GM_ROW_SPLIT_ROWS || maxK < 512 || scratch == nullptr) return gm_gemm<AT, BT>(g, st);
  const long tiles = (long)((maxM + 63) / 64) * ((maxN + 63) / 64) * g.nprob;
  long S = 1;  // a power of two: equal slices, whole K steps
  while (S < GM_ROW_SPLITS && tiles * S < 320 && maxK / (2 * S) >= 128) S *= 2;
  if (S < 2) return gm_gemm<AT, BT>(g, st);
  GEpiArgs ea = {};
  size_t off_f = 0;
  for (int i = 0; i < g.nprob; ++i) {
    GProb& p = g.p[i];
    if (p.accumulate || p.ones_n >= 0 || p.cb) return OSA_EINVAL;  // (weight-gradient problems never come here)
    int Si = (int)S;
    while (Si > 1 && p.K / Si < 128) Si /= 2;
    const size_t need = (size_t)Si * p.M * p.ldc;
    if (off_f + need > scratch_floats) return gm_gemm<AT, BT>(g, st);
    GEpi& e = ea.e[ea.n++];
    e.slab = scratch + off_f; e.C = p.C; e.bias = p.bias; e.aux = p.aux; e.M = p.M; e.N = p.N; e.ldc = p.ldc;
    e.ldaux = p.ldaux; e.act = p.act; e.dact = p.dact; e.S = Si; e.slab_stride = (long)p.M * p.ldc;
    p.C = scratch + off_f; p.slab_stride = e.slab_stride; p.splits = Si;
    p.bias = nullptr; p.aux = nullptr; p.act = -1; p.dact = -1;
    off_f += (need + 3) / 4 * 4;
  }
  g.splits = (int)S;
  const int rc = gm_gemm<AT, BT>(g, st);
  if (rc != OSA_OK) return rc;
  hipLaunchKernelGGL(gm_rows_epilogue_kernel, dim3((unsigned)(((long)maxM * maxN + 255) / 256), ea.n), dim3(256), 0, st, ea);
  return hipGetLastError() == hipSuccess ? OSA_OK : OSA_EHIP;
}

GProb gm_prob() {
  GProb p = {};
  p.splits = 1;
  p.act = -1;
  p.dact = -1;
  p.ones_n = -1;
  return p;
}

// ------------------------------------------------------------------------------------------------
// elementwise / reduction kernels around the GEMMs
// ------------------------------------------------------------------------------------------------
// rows of the minibatch gathered into contiguous, 16-byte aligned, zero-padded rows (idx == nullptr: rows 0..R-1)
__global__ __launch_bounds__(256) void gm_gather_kernel(
    long R, const long* __restrict__ idx, const float* __restrict__ obs, int ld_obs, int obs_dim,
    const float* __restrict__ act, int ld_act, int act_dim, const float* __restrict__ s0, const float* __restrict__ s1,
    const float* __restrict__ s2, const float* __restrict__ s3, const float* __restrict__ s4, float* __restrict__ xg,
    int ldx, float* __restrict__ actg, int lda, float* __restrict__ scal) {
  const long b = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int t = threadIdx.x & 63;
  if (b >= R) return;
  const long row = idx ? idx[b] : b;
  for (int c = t; c < ldx; c += 64) xg[b * ldx + c] = c < obs_dim ? obs[row * ld_obs + c] : 0.f;
  if (act && actg)
    for (int c = t; c < lda; c += 64) actg[b * lda + c] = c < act_dim ? act[row * ld_act + c] : 0.f;
  if (scal && t == 0) {
    if (s0) scal[0 * R + b] = s0[row];
    if (s1) scal[1 * R + b] = s1[row];
    if (s2) scal[2 * R + b] = s2[row];
    if (s3) scal[3 * R + b] = s3[row];
    if (s4) scal[4 * R + b] = s4[row];
  }
}

__global__ __launch_bounds__(256) void gm_loss_kernel(GLossArgs a) {
  __shared__ float red[4];
  const int net = blockIdx.y;
  if (!((a.nets_mask >> net) & 1)) return;
  const long b = (long)blockIdx.x * 256 + threadIdx.x;
  const long row = b < a.R ? (a.idx ? a.idx[b] : b) : 0;
  const GLossMem acc{a, b, row, a.out[net] + b * a.ldo[net], a.dz[net] + b * a.ldz[net]};
  gm_loss_body<false>(a, net, b, blockIdx.x, red, a.ldz[net], true, acc, nullptr, row);
}

struct GRedArgs {
  int P, nblk, nb, act_dim, lda;
  int Pn[3], oLS[3];
  const float* params;
  float* grads;
  const float* ws;      // workspace base: the slab regions
  int L[3];
  int oW[3][GM_MAXL];   // first parameter of layer l; its block (weights + bias) ends where the next one begins
  int S[3][GM_MAXL];
  long slab[3][GM_MAXL];
  const float* dls;     // [nblk][lda]
  const float* lpart;   // [3][nblk][4]
  float* npart;         // [3][nb][2]
  float* stats;         // step statistics [16]
  float entropy_coef, critic_norm_coef;
  int use_critic_norm, nets_mask, loss_kind;
  long R;
};

// grads[net][e] = sum_s slab[net][s][e] (+ log_std: block partials - entropy term; critics: + 2 c w), partial squared
// norms per 1024 elements; block 0 of every network also folds the loss statistics.   grid (nb, 3)
__global__ __launch_bounds__(256) void gm_reduce_kernel(GRedArgs a) {
  __shared__ float red[4];
  const int net = blockIdx.y;
  if (!((a.nets_mask >> net) & 1)) return;
  const bool critic = net != 0;
  const float* __restrict__ p = a.params + (long)net * a.P;
  float gsq = 0.f, psq = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = blockIdx.x * 1024 + q * 256 + threadIdx.x;
    if (e >= a.P) continue;
    float g = 0.f;
    if (e < a.oLS[net]) {
      int l = 0;  // the layer of this parameter (block-uniform except at a handful of boundaries)
      for (int q2 = 1; q2 < a.L[net]; ++q2)
        if (e >= a.oW[net][q2]) l = q2;
      const int size = (l + 1 < a.L[net] ? a.oW[net][l + 1] : a.oLS[net]) - a.oW[net][l];
      const float* s = a.ws + a.slab[net][l] + (e - a.oW[net][l]);
      for (int k = 0; k < a.S[net][l]; ++k) g += s[(long)k * size];
      const float w = p[e];
      if (critic) {
        if (a.use_critic_norm) g += 2.f * a.critic_norm_coef * w;
        psq += w * w;
      }
    } else if (!critic && e < a.oLS[0] + a.act_dim && a.loss_kind != 2) {
      const int d = e - a.oLS[0];
      for (int k = 0; k < a.nblk; ++k) g += a.dls[(long)k * a.lda + d];
      g -= a.entropy_coef / (float)a.act_dim;
    }
    a.grads[(long)net * a.P + e] = g;
    gsq += g * g;
  }
  gsq = gm_block_sum(gsq, red);
  psq = gm_block_sum(psq, red);
  if (threadIdx.x == 0) {
    a.npart[((long)net * a.nb + blockIdx.x) * 2 + 0] = gsq;
    a.npart[((long)net * a.nb + blockIdx.x) * 2 + 1] = psq;
  }
  if (blockIdx.x == 0 && a.stats && a.loss_kind != 2) {
    float l = 0.f, r = 0.f;
    for (int k = threadIdx.x; k < a.nblk; k += 256) {
      l += a.lpart[((long)net * a.nblk + k) * 4 + 0];
      r += a.lpart[((long)net * a.nblk + k) * 4 + 1];
    }
    l = gm_block_sum(l, red);
    r = gm_block_sum(r, red);
    if (threadIdx.x == 0) {
      const float invB = 1.f / (float)a.R;
      if (net == 0) {
        float ent = 0.f;
        for (int d = 0; d < a.act_dim; ++d) ent += 1.41893853320467274178f + p[a.oLS[0] + d];
        ent /= (float)a.act_dim;
        a.stats[2] = l * invB - a.entropy_coef * ent;
        a.stats[3] = r * invB;
        a.stats[4] = ent;
      } else {
        a.stats[net - 1] = l * invB;
      }
    }
  }
}

struct GFinArgs {
  int nb, nets_mask, mode;  // mode 0 clip + Adam, 1 clip only, 2 raw, 3 Adam only (gradients clipped and averaged already)
  const float* npart;
  float* fin;               // [3][8]: {clip factor, step_size, inv_bc2_sqrt, lr}
  int* adam_step;
  float* stats;
  float max_grad_norm, lr_actor, lr_critic, beta1, beta2;
  int use_max_grad_norm;
  const float* lr_dev;
};

// one workgroup per network: total norm in block order, clip factor, Adam's bias corrections of THIS step (float64
// like torch), step counter advanced.   grid (3)
__global__ __launch_bounds__(256) void gm_final_kernel(GFinArgs a) {
  __shared__ float red[4];
  const int net = blockIdx.x;
  if (!((a.nets_mask >> net) & 1)) return;
  float coef = 1.f;
  if (a.mode != 3) {
    float g = 0.f, q = 0.f;
    // (a strided walk per thread, then the fixed-order block sum: deterministic)
    for (int k = threadIdx.x; k < a.nb; k += 256) {
      g += a.npart[((long)net * a.nb + k) * 2 + 0];
      q += a.npart[((long)net * a.nb + k) * 2 + 1];
    }
    g = gm_block_sum(g, red);
    q = gm_block_sum(q, red);
    const float total_norm = sqrtf(g);
    if (threadIdx.x == 0 && a.stats) {
      a.stats[7 + net] = total_norm;
      if (net != 0) a.stats[4 + net] = q;
    }
    if (a.use_max_grad_norm && a.mode != 2) {
      coef = a.max_grad_norm / (total_norm + 1e-6f);
      coef = coef > 1.f ? 1.f : coef;
    }
  }
  if (threadIdx.x == 0) {
    float* f = a.fin + net * 8;
    f[0] = coef;
    if (a.mode == 0 || a.mode == 3) {
      const int step = a.adam_step[net] + 1;
      const bool critic = net != 0;
      const float lr = a.lr_dev ? a.lr_dev[critic ? 1 : 0] : (critic ? a.lr_critic : a.lr_actor);
      f[1] = (float)((double)lr / (1.0 - pow((double)a.beta1, (double)step)));
      f[2] = (float)(1.0 / sqrt(1.0 - pow((double)a.beta2, (double)step)));
      a.adam_step[net] = step;
    }
  }
}

// grid (ceil(P / 1024), 3): clip (+ Adam).  torch.optim.Adam single-tensor step (mlp_device.h osa_adam_update4)
__global__ __launch_bounds__(256) void gm_apply_kernel(int P, int nets_mask, int mode, const float* __restrict__ fin,
                                                       float* __restrict__ params, float* __restrict__ adam_m,
                                                       float* __restrict__ adam_v, float* __restrict__ grads,
                                                       float beta1, float beta2, float eps) {
  const int net = blockIdx.y;
  if (!((nets_mask >> net) & 1)) return;
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= P) return;
  const float coef = fin[net * 8 + 0];
  const long o = (long)net * P + e;
  f32x4 g = *reinterpret_cast<f32x4*>(grads + o);
  g = g * coef;
  if (mode == 1) {
    *reinterpret_cast<f32x4*>(grads + o) = g;
    return;
  }
  f32x4 m = *reinterpret_cast<f32x4*>(adam_m + o), v = *reinterpret_cast<f32x4*>(adam_v + o);
  const f32x4 w = *reinterpret_cast<f32x4*>(params + o);
  const f32x4 wn = osa_adam_update4(g, m, v, w, beta1, beta2, fin[net * 8 + 1], fin[net * 8 + 2], eps);
  *reinterpret_cast<f32x4*>(params + o) = wn;
  *reinterpret_cast<f32x4*>(adam_m + o) = m;
  *reinterpret_cast<f32x4*>(adam_v + o) = v;
}

// rollout: sample the action from the mean rows, log-probability, ActionScale, critics' values
__global__ __launch_bounds__(256) void gm_sample_kernel(
    long N, int act_dim, const float* __restrict__ mean, int ldm, const float* __restrict__ vr, int ldvr,
    const float* __restrict__ vc, int ldvc, const float* __restrict__ log_std, const float* __restrict__ eps,
    unsigned long long seed, unsigned long long offset, const unsigned long long* __restrict__ offset_base,
    int deterministic, int nets_mask, float* __restrict__ act, int ld_act, float* __restrict__ value_r,
    float* __restrict__ value_c, float* __restrict__ logp, float* __restrict__ mean_out, int ld_mean,
    float* __restrict__ act_env, int ld_env, const float* __restrict__ old_min, const float* __restrict__ old_max,
    float min_a, float max_a) {
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  if (row >= N) return;
  if (offset_base) offset += *offset_base;
  if (nets_mask & 1) {
    float lp = 0.f;
    for (int d = 0; d < act_dim; ++d) {
      const float mu = mean[row * ldm + d];
      const float sd = expf(log_std[d]);
      float a = mu;
      if (!deterministic) {
        float e;
        if (eps != nullptr) {
          e = eps[row * act_dim + d];
        } else {  // the same counter mapping as osa_policy_step_kernel: one Philox block per (row, dimension)
          uint32_t w[4];
          osa_philox(seed, offset, (unsigned long long)row * act_dim + d, w);
          float e1;
          osa_box_muller(w[0], w[1], e, e1);
        }
        a = mu + e * sd;
      }
      if (act) act[row * ld_act + d] = a;
      if (act_env) act_env[row * ld_env + d] = osa_action_scale1(a, old_min[d], old_max[d], min_a, max_a);
      if (mean_out) mean_out[row * ld_mean + d] = mu;
      const float z = a - mu;
      lp += -(z * z) / (2.f * (sd * sd)) - logf(sd) - 0.91893853320467274178f;
    }
    if (logp) logp[row] = lp;
  }
  if ((nets_mask & 2) && value_r) value_r[row] = vr[row * ldvr];
  if ((nets_mask & 4) && value_c) value_c[row] = vc[row * ldvc];
}

// full-batch actor statistics from the mean rows: KL(old || new) sums and the line-search sums of osa_actor_eval
__global__ __launch_bounds__(256) void gm_rowstat_kernel(
    long M, int act_dim, const float* __restrict__ mean, int ldm, const float* __restrict__ log_std,
    const float* __restrict__ old_mean, int ld_old, const float* __restrict__ old_log_std,
    const float* __restrict__ act, int ld_act, const float* __restrict__ logp, const float* __restrict__ adv_r,
    const float* __restrict__ adv_c, const float* __restrict__ lagrange, float* __restrict__ mean_out, int ld_mean,
    double* __restrict__ ws) {
  __shared__ double red[17];
  const float lam = lagrange ? *lagrange : 0.f;
  double s_sur = 0.0, s_cost = 0.0, s_kl = 0.0, s_ratio = 0.0;
  for (long row = (long)blockIdx.x * 256 + threadIdx.x; row < M; row += (long)gridDim.x * 256) {
    float lp = 0.f, klp = 0.f;
    for (int d = 0; d < act_dim; ++d) {
      const float mu = mean[row * ldm + d];
      if (mean_out) mean_out[row * ld_mean + d] = mu;
      if (old_mean) {
        const float qs = expf(log_std[d]), ps = expf(old_log_std[d]);
        const float vr = (ps / qs) * (ps / qs);
        const float t1 = (old_mean[row * ld_old + d] - mu) / qs;
        klp += 0.5f * (vr + t1 * t1 - 1.f - logf(vr));
        if (act) {
          const float z = act[row * ld_act + d] - mu;
          lp += -(z * z) / (2.f * (qs * qs)) - logf(qs) - 0.91893853320467274178f;
        }
      }
    }
    s_kl += (double)klp;
    if (act && old_mean) {
      const float ratio = expf(lp - logp[row]);
      const float adv = (adv_r[row] - lam * adv_c[row]) / (1.f + lam);
      s_sur += (double)(ratio * adv);
      s_cost += (double)(ratio * adv_c[row]);
      s_ratio += (double)ratio;
    }
  }
  s_sur = osa_block_sum<256>(s_sur, red);
  s_cost = osa_block_sum<256>(s_cost, red);
  s_kl = osa_block_sum<256>(s_kl, red);
  s_ratio = osa_block_sum<256>(s_ratio, red);
  if (threadIdx.x == 0 && ws) {
    ws[4 * blockIdx.x + 0] = s_sur;
    ws[4 * blockIdx.x + 1] = s_cost;
    ws[4 * blockIdx.x + 2] = s_kl;
    ws[4 * blockIdx.x + 3] = s_ratio;
  }
}

// mode 0: *out = kl_sum / denom (osa_actor_kl);  1: the four outputs of osa_actor_eval
__global__ __launch_bounds__(256) void gm_rowstat_final_kernel(const double* __restrict__ ws, int nblk, int mode,
                                                               double M, double act_dim, double denom,
                                                               float* __restrict__ out) {
  __shared__ double red[17];
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (int k = threadIdx.x; k < nblk; k += 256)
    for (int q = 0; q < 4; ++q) s[q] += ws[4 * k + q];
  for (int q = 0; q < 4; ++q) s[q] = osa_block_sum<256>(s[q], red);
  if (threadIdx.x == 0) {
    if (mode == 0) {
      out[0] = (float)(s[2] / denom);
    } else {
      out[0] = (float)(-s[0] / M);
      out[1] = (float)(s[1] / M);
      out[2] = (float)(s[2] / (M * act_dim));
      out[3] = (float)(s[3] / M);
    }
  }
}


// ------------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------------
// forward of the networks in `mask` over R gathered rows: layer by layer, the networks of a layer in one launch
int gm_forward(const GLayout& lo, const GWs& w, float* ws, const float* params, long R, int mask, hipStream_t st) {
  int maxL = 0;
  for (int net = 0; net < 3; ++net)
    if (((mask >> net) & 1) && lo.n[net].L > maxL) maxL = lo.n[net].L;
  for (int l = 0; l < maxL; ++l) {
    GArgs g = {};
    g.splits = 1;
    for (int net = 0; net < 3; ++net) {
      const GNet& n = lo.n[net];
      if (!((mask >> net) & 1) || l >= n.L) continue;
      GProb p = gm_prob();
      const float* pn = params + (long)net * lo.P;
      p.A = l == 0 ? ws + w.xg : ws + w.h[net][l - 1];
      p.lda = l == 0 ? lo.ldx : n.ldh[l - 1];
      p.B = pn + n.oW[l];
      p.ldb = n.ld[l];
      p.C = ws + w.h[net][l];
      p.ldc = n.ldh[l];
      p.M = (int)R;
      p.N = n.out[l];
      p.K = n.in[l];
      p.bias = pn + n.ob[l];
      p.act = l + 1 < n.L ? n.act : -1;
      g.p[g.nprob++] = p;
    }
    const int rc = gm_gemm_rows<false, false>(g, w.rsl_floats ? ws + w.rsl : nullptr, w.rsl_floats, st);
    if (rc != OSA_OK) return rc;
  }
  return OSA_OK;
}

int gm_gather(const GLayout& lo, const GWs& w, float* ws, long R, const long* idx, const float* obs, int ld_obs,
              const float* act, int ld_act, const float* s0, const float* s1, const float* s2, const float* s3,
              const float* s4, hipStream_t st) {
  hipLaunchKernelGGL(gm_gather_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, R, idx, obs, ld_obs, lo.obs_dim,
                     act, ld_act, lo.act_dim, s0, s1, s2, s3, s4, ws + w.xg, lo.ldx, act ? ws + w.actg : nullptr,
                     lo.lda, ws + w.scal);
  return hipGetLastError() == hipSuccess ? OSA_OK : OSA_EHIP;
}


void gm_fill_ext(GLossArgs& la, const osa_surrogate_ext* ext, const long* idx, float* stats) {
  la.ext_on = 0;
  if (!ext) return;
  la.ext_on = 1; la.idx = idx; la.old_mean = ext->old_mean; la.ld_old_mean = ext->ld_old_mean;
  la.old_log_std = ext->old_log_std; la.ext_kl_coef = ext->kl_coef; la.ext_mask_eta = ext->kl_mask_eta;
  la.ext_ratio_scale = ext->ratio_scale; la.ext_cost_kappa = ext->cost_kappa; la.ext_cost_excess = ext->cost_excess;
  la.stats = stats;
}

// the one-dimensional grid of a skinny launch: tiles of 16 output columns, problem after problem (GSArgs.tile0)
unsigned gs_grid(GSArgs& g, bool at_least_one) {
  int t = 0;
  for (int y = 0; y < g.nprob; ++y) {
    g.tile0[y] = t;
    int n = (g.p[y].ldy + 15) / 16;
    if (at_least_one && n < 1) n = 1;
    t += n;
  }
  for (int y = g.nprob; y <= GS_MAXPROB; ++y) g.tile0[y] = t;
  return (unsigned)t;
}

bool gs_enabled() {  // (A/B and test switch: OSA_GMLP_SKINNY=0 keeps small minibatches on the tiled GEMM)
  const char* v = getenv("OSA_GMLP_SKINNY");
  return !(v != nullptr && v[0] == '0' && v[1] == 0);
}

// One optimiser step (or its gradient: mode 1 / 2) of a minibatch of B <= 64 rows on the skinny kernels: L forward
// launches (layer 0 reads the minibatch's rows in place: no gather), loss + top layer's backward in one launch, L - 2
// more backward-data launches, ONE weight-gradient launch over all layers (norm partials), then clip + Adam from
// recomputed tiles (mode 0: 2 L + 1 launches, 7 for two hidden layers) or gm_final_kernel + the scaling of the written
// gradient (mode 1).
int gm_minibatch_skinny(const GLayout& lo, const GWs& w, float* params, float* adam_m, float* adam_v, int* adam_step,
                        float* grads, const float* obs, int ld_obs, const float* act, int ld_act, const float* logp,
                        const float* target_value_r, const float* target_value_c, const float* adv_r,
                        const float* adv_c, const long* idx, long B, const float* lagrange, const osa_ppo_hparams* hp,
                        int loss_kind, int mode, int mask, float* ws, float* step_stats, const osa_surrogate_ext* ext,
                        hipStream_t st) {
  int maxL = 0;
  // (round 6) mode 0: the clip norm comes out of the forward / top / backward launches (gs_gram_norm) and the top layer's
  // forward pass rides in the launch below it -- 2 L - 1 launches instead of 2 L + 1.  Every active network needs a
  // hidden layer and a top layer of at most 32 outputs; OSA_GMLP_NORM_FUSE=0 keeps round 5's launches (A/B, tests).
  const char* nf = getenv("OSA_GMLP_NORM_FUSE");  // (read per call: the tests switch it inside one process)
  bool fusedn = mode == 0 && !(nf != nullptr && nf[0] == '0' && nf[1] == 0);
  for (int net = 0; net < 3; ++net) {
    if (!((mask >> net) & 1)) continue;
    const GNet& n = lo.n[net];
    if (n.L > maxL) maxL = n.L;
    if (n.L < 2 || n.ldh[n.L - 1] > 32) fusedn = false;
  }
  {
    const char* v = getenv("OSA_GMLP_TOP_FUSE");
    if (v != nullptr && v[0] == '0' && v[1] == 0) fusedn = false;
  }
  const float c2 = hp->use_critic_norm ? 2.f * hp->critic_norm_coef : 0.f;
  auto top_in_fwd = [&](const GNet& n) { return fusedn && n.ldh[n.L - 1] <= 8; };  // the top layer's forward pass fused
  // ---- forward, layer by layer (the networks of a layer in one launch).  No gather launch: layer 0 reads the caller's
  // observation rows through the minibatch's indices, the loss its scalars and action rows likewise
  for (int l = 0; l < maxL; ++l) {
    GSArgs g = {};
    g.R = (int)B;
    g.xrows = l == 0;
    g.xidx = idx;
    g.fused = fusedn;
    int maxN = 0;
    for (int net = 0; net < 3; ++net) {
      const GNet& n = lo.n[net];
      if (!((mask >> net) & 1) || l >= n.L) continue;
      if (l == n.L - 1 && top_in_fwd(n)) continue;  // (its outputs: partial slabs of the launch below)
      GSProb& p = g.p[g.nprob++];
      const float* pn = params + (long)net * lo.P;
      p.X = l == 0 ? obs : ws + w.h[net][l - 1];
      p.ldx = l == 0 ? ld_obs : n.ldh[l - 1];
      p.W = pn + n.oW[l]; p.ldw = n.ld[l];
      p.bias = pn + n.ob[l];
      p.Y = ws + w.h[net][l]; p.ldy = n.ldh[l];
      p.N = n.out[l]; p.K = n.in[l];
      p.act = l + 1 < n.L ? n.act : -1;
      p.net = net;
      if (p.N > maxN) maxN = p.N;
      if (fusedn && l + 1 < n.L) {
        if (net != 0 && c2 != 0.f) p.Ylin = ws + w.ylin[net][l];
        p.slot = ws + w.slots + ((size_t)net * w.slot_stride + w.fs[net][l]) * 4;
        if (l == n.L - 2 && top_in_fwd(n)) {
          p.W2 = pn + n.oW[l + 1]; p.ldw2 = n.ld[l + 1]; p.N2 = n.out[l + 1]; p.ldo2 = n.ldh[l + 1];
          p.oslab = ws + w.oslab[net];
        }
        // the Gram matrix of this layer's input rows, for the norm of its weight gradient
        GSProb& q = g.p[g.nprob++];
        q.X = p.X; q.ldx = p.ldx; q.W = p.X; q.ldw = p.ldx; q.wrows = l == 0;
        q.Y = ws + w.gram[net][l]; q.ldy = 64; q.N = (int)B; q.K = n.in[l]; q.act = -1; q.net = net;
        if (64 > maxN) maxN = 64;
      }
    }
    if (g.nprob == 0) continue;
    if (fusedn && l == 0) {  // Adam's bias corrections of this step, the step counters
      g.fin = ws + w.fin; g.adam_step = adam_step; g.lr_dev = hp->lr_device; g.lr_actor = hp->lr_actor;
      g.lr_critic = hp->lr_critic; g.beta1 = hp->beta1; g.beta2 = hp->beta2; g.fin_mask = mask;
      g.sp[0] = logp; g.sp[1] = adv_r; g.sp[2] = adv_c; g.sp[3] = target_value_r; g.sp[4] = target_value_c;
      g.act = act; g.ld_act = ld_act; g.act_dim = lo.act_dim; g.lda = lo.lda;
      g.scal = ws + w.scal; g.actg = ws + w.actg;
    }
    const unsigned nwg = gs_grid(g, false) + (fusedn && l == 0 ? 1 : 0);  // (+ the auxiliary workgroup)
    if (l == 0)
      hipLaunchKernelGGL(gs_fwd_kernel<true>, dim3(nwg), dim3(64 * GS_WAVES), 0, st, g);
    else
      hipLaunchKernelGGL(gs_fwd_kernel<false>, dim3(nwg), dim3(64 * GS_WAVES), 0, st, g);
  }
  // ---- loss and dL/d(output)
  const GNet& an = lo.n[0];
  GLossArgs la = {};
  la.R = B; la.act_dim = lo.act_dim; la.lda = lo.lda;
  bool fuse = true;  // loss inside the top layer's backward launch: top layers up to 32 wide
  for (int net = 0; net < 3; ++net) {
    const GNet& n = lo.n[net];
    la.out[net] = ws + w.h[net][n.L - 1];
    la.ldo[net] = n.ldh[n.L - 1];
    la.dz[net] = ws + w.zs[net][n.L - 1];
    la.ldz[net] = n.ldh[n.L - 1];
    if (((mask >> net) & 1) && n.ldh[n.L - 1] > 32) fuse = false;
  }
  la.log_std = params + an.oLS; la.lagrange = lagrange;
  la.clip = hp->clip; la.loss_kind = loss_kind; la.nets_mask = mask;
  la.dls = ws + w.dls; la.lpart = ws + w.lpart; la.nblk = w.nblk;
  la.direct = 1; la.ld_act = ld_act; la.act = act; la.idx = idx;
  la.scal = ws + w.scal; la.actg = ws + w.actg;  // (fused: gathered by the first launch)
  la.sp[0] = logp; la.sp[1] = adv_r; la.sp[2] = adv_c; la.sp[3] = target_value_r; la.sp[4] = target_value_c;
  gm_fill_ext(la, ext, idx, step_stats);
  static const bool fuse_on = [] {  // (A/B switch: OSA_GMLP_TOP_FUSE=0 keeps the loss a launch of its own)
    const char* v = getenv("OSA_GMLP_TOP_FUSE");
    return !(v != nullptr && v[0] == '0' && v[1] == 0);
  }();
  fuse = fuse && fuse_on;
  if (fuse) {
    GSArgs g = {};
    g.R = (int)B;
    int maxld = 0;
    for (int net = 0; net < 3; ++net) {
      const GNet& n = lo.n[net];
      if (!((mask >> net) & 1)) continue;
      const int l = n.L - 1;
      GSProb& p = g.p[g.nprob++];
      p.net = net;
      p.W = params + (long)net * lo.P + n.oW[l]; p.ldw = n.ld[l];
      p.N = n.out[l];
      if (l >= 1) {  // (a network without a hidden layer takes part with zero output columns: loss only)
        p.Y = ws + w.zs[net][l - 1]; p.ldy = n.ldh[l - 1];
        p.aux = ws + w.h[net][l - 1]; p.ldaux = n.ldh[l - 1];
        p.K = n.in[l];
        p.act = n.act;
        if (p.ldy > maxld) maxld = p.ldy;
      }
      if (fusedn) {  // (l >= 1 for every active network)
        const float* pn = params + (long)net * lo.P;
        float* sl = ws + w.slots + (size_t)net * w.slot_stride * 4;
        p.c2 = net != 0 ? c2 : 0.f;
        p.btop = pn + n.ob[l];
        p.tslot = sl + (size_t)w.ts[net] * 4;
        p.nslot = sl + (size_t)w.ds[net][l - 1] * 4;
        p.G = ws + w.gram[net][l - 1];
        p.Zlin = net != 0 && c2 != 0.f ? ws + w.ylin[net][l - 1] : nullptr;
        p.bvec = pn + n.ob[l - 1];
        if (top_in_fwd(n)) {
          p.oslab = ws + w.oslab[net]; p.ldo2 = n.ldh[l]; p.nbo = (n.ldh[l - 1] + 15) / 16;
        }
      }
    }
    g.fused = fusedn;
    GSTopFin tf = {};
    tf.stats = step_stats; tf.entropy_coef = hp->entropy_coef; tf.loss_kind = loss_kind; tf.act_dim = lo.act_dim;
    if (g.nprob > 0) hipLaunchKernelGGL(gs_top_kernel, dim3(gs_grid(g, true)), dim3(256), 0, st, g, la, tf);
  } else {
    hipLaunchKernelGGL(gm_loss_kernel, dim3(w.nblk, 3), dim3(256), 0, st, la);
  }
  // ---- backward-data, from the top of EACH network
  for (int step = fuse ? 1 : 0; step + 1 < maxL; ++step) {
    GSArgs g = {};
    g.R = (int)B;
    int maxK = 0;
    for (int net = 0; net < 3; ++net) {
      const GNet& n = lo.n[net];
      const int l = n.L - 1 - step;
      if (!((mask >> net) & 1) || l < 1) continue;
      GSProb& p = g.p[g.nprob++];
      p.X = ws + w.zs[net][l]; p.ldx = n.ldh[l];
      p.W = params + (long)net * lo.P + n.oW[l]; p.ldw = n.ld[l];
      p.Y = ws + w.zs[net][l - 1]; p.ldy = n.ldh[l - 1];
      p.aux = ws + w.h[net][l - 1]; p.ldaux = n.ldh[l - 1];
      p.N = n.out[l]; p.K = n.in[l];
      p.act = n.act;
      p.net = net;
      if (p.K > maxK) maxK = p.K;
      if (fusedn) {  // the norm of layer l - 1's weight gradient, whose dZ this launch produces
        const float* pn = params + (long)net * lo.P;
        p.c2 = net != 0 ? c2 : 0.f;
        p.nslot = ws + w.slots + ((size_t)net * w.slot_stride + w.ds[net][l - 1]) * 4;
        p.G = ws + w.gram[net][l - 1];
        p.Zlin = net != 0 && c2 != 0.f ? ws + w.ylin[net][l - 1] : nullptr;
        p.bvec = pn + n.ob[l - 1];
      }
    }
    g.fused = fusedn;
    if (g.nprob > 0) hipLaunchKernelGGL(gs_bwd_kernel, dim3(gs_grid(g, false)), dim3(64 * GS_WAVES), 0, st, g);
  }
  // ---- weight gradients of all layers: norm partials (+ the gradient itself for modes 1 / 2)
  GSWArgs wa = {};
  for (int net = 0; net < 3; ++net) {
    const GNet& n = lo.n[net];
    wa.L[net] = n.L; wa.ntile[net] = w.sk_ntile[net]; wa.oLS[net] = n.oLS;
    for (int l = 0; l < n.L; ++l) {
      GSWLayer& y = wa.l[net][l];
      y.dZ = ws + w.zs[net][l]; y.ldz = n.ldh[l];
      y.H = l == 0 ? obs : ws + w.h[net][l - 1];
      y.ldh = l == 0 ? ld_obs : n.ldh[l - 1];
      y.hrows = l == 0; y.hidx = idx;
      y.out = n.out[l]; y.in = n.in[l]; y.ldw = n.ld[l]; y.oW = n.oW[l]; y.ob = n.ob[l];
      y.tile0 = w.sk_tile0[net][l]; y.tk = w.sk_tk[net][l];
    }
  }
  wa.R = (int)B; wa.P = lo.P; wa.maxT1 = w.sk_maxT1; wa.act_dim = lo.act_dim; wa.lda = lo.lda; wa.nblk = w.nblk;
  wa.nets_mask = mask; wa.loss_kind = loss_kind; wa.write_grads = mode != 0; wa.use_critic_norm = hp->use_critic_norm;
  wa.params = params; wa.adam_m = adam_m; wa.adam_v = adam_v; wa.grads = grads; wa.npart = ws + w.npart;
  wa.fin = ws + w.fin; wa.dls = ws + w.dls; wa.lpart = ws + w.lpart; wa.stats = step_stats;
  wa.entropy_coef = hp->entropy_coef; wa.critic_norm_coef = hp->critic_norm_coef; wa.beta1 = hp->beta1;
  wa.beta2 = hp->beta2; wa.eps = hp->adam_eps;
  wa.fold = mode == 0; wa.use_max_grad_norm = hp->use_max_grad_norm; wa.max_grad_norm = hp->max_grad_norm;
  wa.adam_step = adam_step; wa.lr_dev = hp->lr_device; wa.lr_actor = hp->lr_actor; wa.lr_critic = hp->lr_critic;
  // (four workgroups per compute unit -- 35 KB of LDS each -- is the best occupancy measured: capped at 3 / 2 / 1 the
  // two launches take 41 / 43 / 62 us instead of 33 at 1024 x 1024, profiles/HISTORY.md)
  if (fusedn) {
    wa.fold = 2; wa.slots = ws + w.slots; wa.slot_stride = w.slot_stride;
    for (int net = 0; net < 3; ++net) wa.nslot[net] = w.nslot[net];
  } else {
    hipLaunchKernelGGL(gs_wgrad_kernel<0>, dim3(w.sk_maxT1, 3), dim3(256), 0, st, wa);
  }
  if (mode == 0) {
    hipLaunchKernelGGL(gs_wgrad_kernel<1>, dim3(w.sk_maxT1, 3), dim3(256), 0, st, wa);
    OSA_CHECK_LAUNCH();
    return OSA_OK;
  }
  GFinArgs fa = {};
  fa.nb = w.sk_maxT1; fa.nets_mask = mask; fa.mode = mode; fa.npart = ws + w.npart; fa.fin = ws + w.fin;
  fa.adam_step = adam_step; fa.stats = step_stats; fa.max_grad_norm = hp->max_grad_norm;
  fa.lr_actor = hp->lr_actor; fa.lr_critic = hp->lr_critic; fa.beta1 = hp->beta1; fa.beta2 = hp->beta2;
  fa.use_max_grad_norm = hp->use_max_grad_norm; fa.lr_dev = hp->lr_device;
  hipLaunchKernelGGL(gm_final_kernel, dim3(3), dim3(256), 0, st, fa);
  if (mode == 1)
    hipLaunchKernelGGL(gm_apply_kernel, dim3((lo.P / 4 + 255) / 256, 3), dim3(256), 0, st, lo.P, mask, mode,
                       ws + w.fin, params, adam_m, adam_v, grads, hp->beta1, hp->beta2, hp->adam_eps);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

}  // namespace

extern "C" {

int osa_gmlp_layout(const osa_gmlp_desc* desc, int* out) {
  OSA_REQUIRE(desc && out);
  GLayout lo;
  const int rc = gm_make_layout(desc, &lo);
  if (rc != OSA_OK) return rc;
  out[0] = lo.P;
  out[1] = lo.n[0].oLS;
  int k = 2;
  for (int net = 0; net < 3; ++net)
    for (int l = 0; l < GM_MAXL; ++l) {
      const bool on = l < lo.n[net].L;
      out[k++] = on ? lo.n[net].oW[l] : -1;
      out[k++] = on ? lo.n[net].ob[l] : -1;
      out[k++] = on ? lo.n[net].ld[l] : 0;
    }
  return OSA_OK;
}

size_t osa_gmlp_ws_floats(const osa_gmlp_desc* desc, long rows) {
  GLayout lo;
  if (!desc || rows < 1 || gm_make_layout(desc, &lo) != OSA_OK) return 0;
  return gm_ws(lo, rows).total;
}

int osa_gmlp_policy_step(const osa_gmlp_desc* desc, const float* params, const float* obs, int ld_obs, long N,
                         const float* eps, unsigned long long seed, unsigned long long offset,
                         const unsigned long long* offset_base, int deterministic, int nets_mask, float* act,
                         int ld_act, float* value_r, float* value_c, float* logp, float* mean_out, int ld_mean,
                         float* act_env, int ld_env, const float* old_min, const float* old_max, float min_action,
                         float max_action, float* ws, size_t ws_floats, void* stream) {
  OSA_REQUIRE(desc && params && obs && ws && N > 0 && ld_obs >= desc->obs_dim);
  GLayout lo;
  int rc = gm_make_layout(desc, &lo);
  if (rc != OSA_OK) return rc;
  const GWs w = gm_ws(lo, N);
  if (ws_floats < w.total) return OSA_EINVAL;
  hipStream_t st = osa_stream(stream);
  if ((rc = gm_gather(lo, w, ws, N, nullptr, obs, ld_obs, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
                      st)) != OSA_OK)
    return rc;
  if ((rc = gm_forward(lo, w, ws, params, N, nets_mask & 7, st)) != OSA_OK) return rc;
  const GNet &a = lo.n[0], &r = lo.n[1], &c = lo.n[2];
  hipLaunchKernelGGL(gm_sample_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, N, lo.act_dim,
                     ws + w.h[0][a.L - 1], a.ldh[a.L - 1], ws + w.h[1][r.L - 1], r.ldh[r.L - 1], ws + w.h[2][c.L - 1],
                     c.ldh[c.L - 1], params + a.oLS, eps, seed, offset, offset_base, deterministic, nets_mask, act,
                     ld_act, value_r, value_c, logp, mean_out, ld_mean, act_env, ld_env, old_min, old_max, min_action,
                     max_action);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_gmlp_minibatch(const osa_gmlp_desc* desc, float* params, float* adam_m, float* adam_v, int* adam_step,
                       float* grads, const float* obs, int ld_obs, const float* act, int ld_act, const float* logp,
                       const float* target_value_r, const float* target_value_c, const float* adv_r,
                       const float* adv_c, const long* idx, long B, const float* lagrange, const osa_ppo_hparams* hp,
                       int loss_kind, int mode, int nets_mask, const float* vec, float fvp_scale, float* ws,
                       size_t ws_floats, float* step_stats, void* stream) {
  return osa_gmlp_minibatch_ext(desc, params, adam_m, adam_v, adam_step, grads, obs, ld_obs, act, ld_act, logp,
                                target_value_r, target_value_c, adv_r, adv_c, idx, B, lagrange, hp, loss_kind, mode,
                                nets_mask, vec, fvp_scale, ws, ws_floats, step_stats, nullptr, stream);
}

int osa_gmlp_minibatch_ext(const osa_gmlp_desc* desc, float* params, float* adam_m, float* adam_v, int* adam_step,
                           float* grads, const float* obs, int ld_obs, const float* act, int ld_act, const float* logp,
                           const float* target_value_r, const float* target_value_c, const float* adv_r,
                           const float* adv_c, const long* idx, long B, const float* lagrange,
                           const osa_ppo_hparams* hp, int loss_kind, int mode, int nets_mask, const float* vec,
                           float fvp_scale, float* ws, size_t ws_floats, float* step_stats,
                           const osa_surrogate_ext* ext, void* stream) {
  OSA_REQUIRE(desc && params && adam_m && adam_v && adam_step && grads && obs && hp && ws && B > 0);
  if (ext) {
    OSA_REQUIRE(loss_kind != 2 && ext->old_mean && ext->old_log_std && ext->ld_old_mean >= desc->act_dim);
    // the trust-mask mean and the penalty are minibatch-level quantities of ONE 256-row block of the loss kernel
    if ((ext->kl_mask_eta >= 0.f || ext->cost_kappa > 0.f) && B > 256) return OSA_EUNSUPPORTED;
  }
  OSA_REQUIRE(mode >= 0 && mode <= 2 && loss_kind >= 0 && loss_kind <= 2);
  OSA_REQUIRE(loss_kind == 2 ? vec != nullptr : (act && logp && target_value_r && target_value_c && adv_r && adv_c));
  GLayout lo;
  int rc = gm_make_layout(desc, &lo);
  if (rc != OSA_OK) return rc;
  const GWs w = gm_ws(lo, B);
  if (ws_floats < w.total) return OSA_EINVAL;
  hipStream_t st = osa_stream(stream);
  int mask = nets_mask & (hp->use_cost ? 7 : 3);
  if (loss_kind == 2) mask = 1;
  if (B <= GS_MAX_ROWS && loss_kind != 2 && gs_enabled())
    return gm_minibatch_skinny(lo, w, params, adam_m, adam_v, adam_step, grads, obs, ld_obs, act, ld_act, logp,
                               target_value_r, target_value_c, adv_r, adv_c, idx, B, lagrange, hp, loss_kind, mode, mask,
                               ws, step_stats, ext, st);
  if ((rc = gm_gather(lo, w, ws, B, idx, obs, ld_obs, loss_kind == 2 ? nullptr : act, ld_act, logp, adv_r, adv_c,
                      target_value_r, target_value_c, st)) != OSA_OK)
    return rc;
  if ((rc = gm_forward(lo, w, ws, params, B, mask, st)) != OSA_OK) return rc;
  const GNet& an = lo.n[0];
  if (loss_kind == 2) {
    // forward-mode tangent of the mean network along `vec` (natural_pg.py:91-119 without a double backward: for a
    // Gaussian policy with state-independent log_std the Hessian of mean KL at theta_old is J^T diag(1/sigma^2) J /
    // (M D_a), profiles/HISTORY.md §3.1):  T_l = (H_{l-1} Vw_l^T + vb_l + T_{l-1} W_l^T) * act'(H_l)
    for (int l = 0; l < an.L; ++l) {
      GArgs g = {};
      g.splits = 1;
      g.nprob = 1;
      GProb p = gm_prob();
      p.A = l == 0 ? ws + w.xg : ws + w.h[0][l - 1];
      p.lda = l == 0 ? lo.ldx : an.ldh[l - 1];
      p.B = vec + an.oW[l];
      p.ldb = an.ld[l];
      p.C = ws + w.t[l];
      p.ldc = an.ldh[l];
      p.M = (int)B; p.N = an.out[l]; p.K = an.in[l];
      p.bias = vec + an.ob[l];
      if (l == 0 && l + 1 < an.L) { p.dact = an.act; p.aux = ws + w.h[0][l]; p.ldaux = an.ldh[l]; }
      g.p[0] = p;
      if ((rc = gm_gemm<false, false>(g, st)) != OSA_OK) return rc;
      if (l > 0) {
        GProb q = gm_prob();
        q.A = ws + w.t[l - 1]; q.lda = an.ldh[l - 1];
        q.B = params + an.oW[l]; q.ldb = an.ld[l];
        q.C = ws + w.t[l]; q.ldc = an.ldh[l];
        q.M = (int)B; q.N = an.out[l]; q.K = an.in[l];
        q.accumulate = 1;
        if (l + 1 < an.L) { q.dact = an.act; q.aux = ws + w.h[0][l]; q.ldaux = an.ldh[l]; }
        g.p[0] = q;
        if ((rc = gm_gemm<false, false>(g, st)) != OSA_OK) return rc;
      }
    }
  }
  // ---- loss and dL/d(output)
  GLossArgs la = {};
  la.R = B; la.act_dim = lo.act_dim; la.lda = lo.lda;
  int cur[3] = {0, 0, 0};  // which ping-pong buffer holds dZ of the layer in flight
  for (int net = 0; net < 3; ++net) {
    const GNet& n = lo.n[net];
    la.out[net] = ws + w.h[net][n.L - 1];
    la.ldo[net] = n.ldh[n.L - 1];
    la.dz[net] = ws + w.z[net][0];
    la.ldz[net] = n.ldh[n.L - 1];
  }
  la.actg = ws + w.actg; la.scal = ws + w.scal; la.log_std = params + an.oLS; la.lagrange = lagrange;
  la.clip = hp->clip; la.loss_kind = loss_kind; la.nets_mask = mask;
  la.dls = ws + w.dls; la.lpart = ws + w.lpart; la.nblk = w.nblk;
  la.tmean = ws + w.t[an.L - 1]; la.ldt = an.ldh[an.L - 1]; la.fvp_scale = fvp_scale;
  gm_fill_ext(la, ext, idx, step_stats);
  hipLaunchKernelGGL(gm_loss_kernel, dim3(w.nblk, 3), dim3(256), 0, st, la);
  OSA_CHECK_LAUNCH();
  // ---- backward: weight (+ bias) gradients into the slabs, then dZ of the layer below
  int maxL = 0;
  for (int net = 0; net < 3; ++net)
    if (((mask >> net) & 1) && lo.n[net].L > maxL) maxL = lo.n[net].L;
  for (int step = 0; step < maxL; ++step) {  // step counts layers from the top of EACH network
    GArgs gw = {}, gd = {};
    gw.splits = 1;
    gd.splits = 1;
    for (int net = 0; net < 3; ++net) {
      const GNet& n = lo.n[net];
      const int l = n.L - 1 - step;
      if (!((mask >> net) & 1) || l < 0) continue;
      float* slab = ws + w.slab[net][l];  // [S_l][out x ld | bias block]
      GProb p = gm_prob();  // dW_l[out][in] (+ db_l) = dZ_l^T [out][rows] . {H_{l-1}, 1}[rows][in + 1]
      p.A = ws + w.z[net][cur[net]]; p.lda = n.ldh[l];
      p.B = l == 0 ? ws + w.xg : ws + w.h[net][l - 1];
      p.ldb = l == 0 ? lo.ldx : n.ldh[l - 1];
      p.C = slab; p.ldc = n.ld[l];
      p.cb = slab + (long)n.out[l] * n.ld[l]; p.cb_pad = r4(n.out[l]);
      p.M = n.out[l]; p.N = n.in[l]; p.K = (int)B;
      p.ones_n = n.in[l];
      p.slab_stride = (long)n.out[l] * n.ld[l] + r4(n.out[l]);
      p.splits = w.S[net][l];
      if (p.splits > gw.splits) gw.splits = p.splits;
      gw.p[gw.nprob++] = p;
      if (l > 0) {  // dZ_{l-1}[rows][in] = (dZ_l[rows][out] . W_l[out][in]) * act'(H_{l-1})
        GProb q = gm_prob();
        q.A = ws + w.z[net][cur[net]]; q.lda = n.ldh[l];
        q.B = params + (long)net * lo.P + n.oW[l]; q.ldb = n.ld[l];
        q.C = ws + w.z[net][cur[net] ^ 1]; q.ldc = n.ldh[l - 1];
        q.M = (int)B; q.N = n.in[l]; q.K = n.out[l];
        q.dact = n.act; q.aux = ws + w.h[net][l - 1]; q.ldaux = n.ldh[l - 1];
        gd.p[gd.nprob++] = q;
      }
    }
    if ((rc = gm_gemm<true, true>(gw, st)) != OSA_OK) return rc;
    if (gd.nprob > 0 && (rc = gm_gemm_rows<false, true>(gd, w.rsl_floats ? ws + w.rsl : nullptr, w.rsl_floats, st)) != OSA_OK)
      return rc;
    for (int net = 0; net < 3; ++net)
      if (((mask >> net) & 1) && lo.n[net].L - 1 - step > 0) cur[net] ^= 1;
  }
  // ---- slab sum, L2 / entropy terms, norms; clip factor and Adam scalars; clip (+ Adam)
  GRedArgs ra = {};
  ra.P = lo.P; ra.nblk = w.nblk; ra.nb = w.nb; ra.act_dim = lo.act_dim; ra.lda = lo.lda;
  for (int net = 0; net < 3; ++net) {
    ra.Pn[net] = lo.n[net].Pn; ra.oLS[net] = lo.n[net].oLS; ra.L[net] = lo.n[net].L;
    for (int l = 0; l < GM_MAXL; ++l) {
      ra.oW[net][l] = l < lo.n[net].L ? lo.n[net].oW[l] : 0;
      ra.S[net][l] = w.S[net][l];
      ra.slab[net][l] = (long)w.slab[net][l];
    }
  }
  ra.params = params; ra.grads = grads; ra.ws = ws; ra.dls = ws + w.dls; ra.lpart = ws + w.lpart;
  ra.npart = ws + w.npart; ra.stats = step_stats; ra.entropy_coef = hp->entropy_coef;
  ra.critic_norm_coef = hp->critic_norm_coef; ra.use_critic_norm = hp->use_critic_norm; ra.nets_mask = mask;
  ra.loss_kind = loss_kind; ra.R = B;
  hipLaunchKernelGGL(gm_reduce_kernel, dim3(w.nb, 3), dim3(256), 0, st, ra);
  GFinArgs fa = {};
  fa.nb = w.nb; fa.nets_mask = mask; fa.mode = mode; fa.npart = ws + w.npart; fa.fin = ws + w.fin;
  fa.adam_step = adam_step; fa.stats = step_stats; fa.max_grad_norm = hp->max_grad_norm;
  fa.lr_actor = hp->lr_actor; fa.lr_critic = hp->lr_critic; fa.beta1 = hp->beta1; fa.beta2 = hp->beta2;
  fa.use_max_grad_norm = hp->use_max_grad_norm; fa.lr_dev = hp->lr_device;
  hipLaunchKernelGGL(gm_final_kernel, dim3(3), dim3(256), 0, st, fa);
  if (mode != 2)
    hipLaunchKernelGGL(gm_apply_kernel, dim3((lo.P / 4 + 255) / 256, 3), dim3(256), 0, st, lo.P, mask, mode,
                       ws + w.fin, params, adam_m, adam_v, grads, hp->beta1, hp->beta2, hp->adam_eps);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_gmlp_adam_apply(const osa_gmlp_desc* desc, float* params, float* adam_m, float* adam_v, int* adam_step,
                        float* grads, const osa_ppo_hparams* hp, int nets_mask, float* fin8x3, void* stream) {
  OSA_REQUIRE(desc && params && adam_m && adam_v && adam_step && grads && hp && fin8x3);
  GLayout lo;
  const int rc = gm_make_layout(desc, &lo);
  if (rc != OSA_OK) return rc;
  hipStream_t st = osa_stream(stream);
  GFinArgs fa = {};
  fa.nb = 0; fa.nets_mask = nets_mask; fa.mode = 3; fa.npart = nullptr; fa.fin = fin8x3; fa.adam_step = adam_step;
  fa.stats = nullptr; fa.lr_actor = hp->lr_actor; fa.lr_critic = hp->lr_critic; fa.beta1 = hp->beta1;
  fa.beta2 = hp->beta2; fa.lr_dev = hp->lr_device;
  hipLaunchKernelGGL(gm_final_kernel, dim3(3), dim3(256), 0, st, fa);
  hipLaunchKernelGGL(gm_apply_kernel, dim3((lo.P / 4 + 255) / 256, 3), dim3(256), 0, st, lo.P, nets_mask, 3, fin8x3,
                     params, adam_m, adam_v, grads, hp->beta1, hp->beta2, hp->adam_eps);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

// osa_actor_kl / osa_actor_eval for general networks: actor forward over M rows + row statistics.
//   kind 0: snapshot (old_mean == NULL: only mean_out) or KL(old || new) -> out[0] (reduce_mode as osa_actor_kl)
//   kind 1: the four line-search sums of osa_actor_eval -> out[0..3]
int osa_gmlp_actor_stats(const osa_gmlp_desc* desc, const float* actor_params, const float* obs, int ld_obs, long M,
                         const float* old_mean, int ld_old, const float* old_log_std, int kind, int reduce_mode,
                         const float* act, int ld_act, const float* logp, const float* adv_r, const float* adv_c,
                         const float* lagrange, float* mean_out, int ld_mean, float* ws, size_t ws_floats, float* out,
                         void* stream) {
  OSA_REQUIRE(desc && actor_params && obs && ws && M > 0);
  OSA_REQUIRE(kind == 0 || (old_mean && old_log_std && act && logp && adv_r && adv_c && out));
  GLayout lo;
  int rc = gm_make_layout(desc, &lo);
  if (rc != OSA_OK) return rc;
  const GWs w = gm_ws(lo, M);
  if (ws_floats < w.total) return OSA_EINVAL;
  hipStream_t st = osa_stream(stream);
  if ((rc = gm_gather(lo, w, ws, M, nullptr, obs, ld_obs, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
                      st)) != OSA_OK)
    return rc;
  // (actor_params points at the actor's block: network 0 of a [3][P] tensor or a candidate vector)
  if ((rc = gm_forward(lo, w, ws, actor_params, M, 1, st)) != OSA_OK) return rc;
  const GNet& an = lo.n[0];
  double* dws = reinterpret_cast<double*>(ws + w.dws);
  int nblk = (int)((M + 255) / 256);
  if (nblk > 1024) nblk = 1024;
  hipLaunchKernelGGL(gm_rowstat_kernel, dim3(nblk), dim3(256), 0, st, M, lo.act_dim, ws + w.h[0][an.L - 1],
                     an.ldh[an.L - 1], actor_params + an.oLS, old_mean, ld_old, old_log_std, kind == 1 ? act : nullptr,
                     ld_act, logp, adv_r, adv_c, lagrange, mean_out, ld_mean, dws);
  if (old_mean && out) {
    const double denom = reduce_mode == 0 ? (double)M : (double)M * lo.act_dim;
    hipLaunchKernelGGL(gm_rowstat_final_kernel, dim3(1), dim3(256), 0, st, dws, nblk, kind, (double)M,
                       (double)lo.act_dim, denom, out);
  }
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

}  // extern "C"
