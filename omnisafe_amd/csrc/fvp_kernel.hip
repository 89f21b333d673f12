// Fisher-vector product of the actor over the whole rollout, throughput-shaped (round 4).
//
// NaturalPG._fvp (omnisafe/algorithms/on_policy/naive/natural_pg.py:91-119) = J^T diag(1 / sigma^2) J v / (M D_a) on the
// mean network (profiles/HISTORY.md §3.1: a JVP through the network, then the ordinary backward pass; no double backward).  The
// general gradient kernel (osa_mb_grad_kernel, mlp_kernels.hip) served it so far: every wave fetched the weights AND
// the vector from L2 for each of its 16 rows' three passes (0.5 MB per 64-row chunk and workgroup) and every chunk
// read-modified-wrote the 33 KB gradient block in global memory -- 163 us per product at 65 536 rows = 16 % of the
// float32-MFMA peak, matrix pipe busy 22 % (profiles/r4_pmc_sq_mfma_busy_TRPOLag.md).
//
// Here one workgroup per compute unit keeps BOTH parameter blocks -- theta and v -- in LDS for the whole launch (rows
// padded by 4 floats: the 16-byte fragment reads of 16 lanes hit 16 distinct bank groups; the S-layout code of
// mlp_device.h reads them through generic pointers), walks through its chunks of 64 rows and accumulates the weight
// gradient IN REGISTERS (a wave owns one 16-row tile of dW1 / dW2 and one 16-column tile of dW3: 9-11 accumulator
// tiles); one slab per workgroup at the end, summed by osa_fvp_reduce_kernel in the order of osa_slab_reduce_kernel.
//
// Same arithmetic as the general kernel, instruction for instruction -- the forward pass is the same osa_mlp_forward,
// the JVP / backward / contraction loops are the same MFMA sequences, a workgroup's chunks are the same
// (blockIdx.x + k * nblk) and are added in the same order -- so the product is BIT-IDENTICAL to the old path
// (tests/test_trust_region_gpu.py::test_fast_fvp_is_bit_identical_to_the_general_kernel) and every reference golden of
// the trust-region family holds unchanged.  Shapes: hidden width 64 (the YAML default), observations up to 80 wide
// (the two padded parameter blocks + the four [64][68] tiles fill 150 - 158 of the 160 KB); everything else keeps the
// general kernel (osa_actor_fvp_raw decides).
#include <stdlib.h>

#include "mlp_device.h"

#define OFV_NSTAT 16  // OSA_NSTAT of mlp_kernels.hip: statistics slots behind the P gradient entries of a slab

// Phase clocks of a chunk (tools/build_variant_lib.sh fvpclocks fvp_kernel.hip -DOFV_CLOCKS; tools/fvp_phase_clocks.py):
// thread 0 of workgroup 0 accumulates the shader cycles between the marks over its chunks.  Not in the product build.
#ifdef OFV_CLOCKS
__device__ long long ofv_clk[16];
#define OFV_MARK(k)                                                         \
  do {                                                                      \
    const long long now_ = (long long)__builtin_amdgcn_s_memtime();         \
    ofv_acc_[k] += now_ - ofv_t_;  /* (registers: a memory update here would wait for every load in flight) */ \
    ofv_t_ = now_;                                                          \
  } while (0)
extern "C" int osa_debug_fvp_clocks(long long* out16, int reset) {
  if (reset) {
    long long z[16] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(ofv_clk), z, sizeof(z)) == hipSuccess ? 0 : 1;
  }
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(ofv_clk), 16 * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#else
#define OFV_MARK(k) do { } while (0)
#endif

struct OsaFvpArgs {
  OsaNet nd;
  const float* params;  // actor block [P]
  const float* vec;     // v, padded actor layout [P]
  const float* obs;
  int ld_obs;
  int M;
  int nblk;
  float* slabs;  // [nblk][P + OFV_NSTAT]
  float* grads;  // [P] (reduce kernel)
  float* stats;
  float fvp_scale;
};

// padded LDS layout of one parameter block (strides + 4 floats; KB and the logical sizes unchanged)
__device__ __forceinline__ OsaNet ofv_padded(const OsaNet& nd) {
  OsaNet nl = nd;
  nl.INP = nd.INP + 4;
  nl.H = nd.H + 4;
  nl.oW1 = 0;
  nl.ob1 = nl.oW1 + nd.H * nl.INP;
  nl.oW2 = nl.ob1 + nd.H;
  nl.ob2 = nl.oW2 + nd.H * nl.H;
  nl.oW3 = nl.ob2 + nd.H;
  nl.ob3 = nl.oW3 + nd.OUTP * nl.H;
  nl.oLS = nl.ob3 + nd.OUTP;
  nl.P = nl.oLS + nd.OUTP;
  return nl;
}

template <int KBT>
__device__ __forceinline__ void ofv_stage(const OsaNet& nd, const OsaNet& nl, const float* __restrict__ src,
                                          float* __restrict__ dst) {
  constexpr int H = 64, INP = 16 * KBT;
  for (int e = threadIdx.x; e < H * INP / 4; e += blockDim.x) {  // 16-byte pieces
    const int r = e / (INP / 4), c = 4 * (e - r * (INP / 4));
    *reinterpret_cast<f32x4*>(dst + nl.oW1 + r * nl.INP + c) = *reinterpret_cast<const f32x4*>(src + nd.oW1 + r * INP + c);
  }
  for (int e = threadIdx.x; e < H * H / 4; e += blockDim.x) {
    const int r = e / (H / 4), c = 4 * (e - r * (H / 4));
    *reinterpret_cast<f32x4*>(dst + nl.oW2 + r * nl.H + c) = *reinterpret_cast<const f32x4*>(src + nd.oW2 + r * H + c);
  }
  for (int e = threadIdx.x; e < nd.OUTP * H / 4; e += blockDim.x) {
    const int r = e / (H / 4), c = 4 * (e - r * (H / 4));
    *reinterpret_cast<f32x4*>(dst + nl.oW3 + r * nl.H + c) = *reinterpret_cast<const f32x4*>(src + nd.oW3 + r * H + c);
  }
  for (int e = threadIdx.x; e < H; e += blockDim.x) {
    dst[nl.ob1 + e] = src[nd.ob1 + e];
    dst[nl.ob2 + e] = src[nd.ob2 + e];
  }
  for (int e = threadIdx.x; e < nd.OUTP; e += blockDim.x) {
    dst[nl.ob3 + e] = src[nd.ob3 + e];
    dst[nl.oLS + e] = src[nd.oLS + e];
  }
}

// Software pipeline of N MFMA groups whose A fragments come from LDS (round 5): group idx + D is requested BEFORE the
// MFMAs of group idx issue, and sched_barriers pin that order.  With one wave per SIMD nothing else hides an LDS round
// trip: left to itself the compiler sinks every ds_read next to its consumer and drains lgkmcnt in front of each MFMA
// group (197 s_waitcnt in the round-4 kernel; 36 % of a chunk's cycles were neither MFMA nor VALU).  The MFMA order is
// untouched: same bits.
#define OFV_SB() __builtin_amdgcn_sched_barrier(0)
template <int N, int D, typename LD, typename MM>
__device__ __forceinline__ void ofv_pipe(LD ld, MM mm) {
  f32x4 buf[D + 1][2];
#pragma unroll
  for (int k = 0; k < D; ++k)
    if (k < N) ld(k, buf[k % (D + 1)]);
#pragma unroll
  for (int idx = 0; idx < N; ++idx) {
    if (idx + D < N) ld(idx + D, buf[(idx + D) % (D + 1)]);
    OFV_SB();
    mm(idx, buf[idx % (D + 1)]);
    OFV_SB();
  }
}
#define OFV_D 2  // groups in flight ahead of the MFMAs (a group = 4 or 8 MFMAs = 128 / 256 cycles)

// osa_mlp_forward (mlp_device.h) on x fragments that are already in registers: the same MFMA sequence (K blocks outer,
// hidden tiles inner, the four k of a fragment in order), weights from the padded LDS block
template <int OT, int KBT>
__device__ __forceinline__ void ofv_forward(const OsaNet& nl, const float* __restrict__ p, const f32x4 (&xf)[KBT], int act,
                                            f32x4 (&h1)[4], f32x4 (&h2)[4], f32x4 (&out)[OT]) {
  constexpr int HT = 4;
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const float* __restrict__ W1 = p + nl.oW1;
  const float* __restrict__ W2 = p + nl.oW2;
  const float* __restrict__ W3 = p + nl.oW3;
#pragma unroll
  for (int t = 0; t < HT; ++t) h1[t] = *reinterpret_cast<const f32x4*>(p + nl.ob1 + 16 * t + 4 * g);
  // (two hidden tiles per group, their MFMAs alternating: no MFMA issues right behind the one whose result it accumulates
  // onto; every accumulator still receives its products in the same order)
  ofv_pipe<KBT * HT / 2, OFV_D>(
      [&](int idx, f32x4 (&d)[2]) {
        const int kb = idx / (HT / 2), t = 2 * (idx % (HT / 2));
        d[0] = *reinterpret_cast<const f32x4*>(W1 + (16 * t + i) * nl.INP + 16 * kb + 4 * g);
        d[1] = *reinterpret_cast<const f32x4*>(W1 + (16 * (t + 1) + i) * nl.INP + 16 * kb + 4 * g);
      },
      [&](int idx, const f32x4 (&w)[2]) {
        const int kb = idx / (HT / 2), t = 2 * (idx % (HT / 2));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          h1[t] = OSA_MFMA(w[0][s], xf[kb][s], h1[t]);
          h1[t + 1] = OSA_MFMA(w[1][s], xf[kb][s], h1[t + 1]);
        }
      });
#pragma unroll
  for (int t = 0; t < HT; ++t) h1[t] = osa_act4(h1[t], act);
#pragma unroll
  for (int t = 0; t < HT; ++t) h2[t] = *reinterpret_cast<const f32x4*>(p + nl.ob2 + 16 * t + 4 * g);
  ofv_pipe<HT * HT / 2, OFV_D>(
      [&](int idx, f32x4 (&d)[2]) {
        const int kb = idx / (HT / 2), t = 2 * (idx % (HT / 2));
        d[0] = *reinterpret_cast<const f32x4*>(W2 + (16 * t + i) * nl.H + 16 * kb + 4 * g);
        d[1] = *reinterpret_cast<const f32x4*>(W2 + (16 * (t + 1) + i) * nl.H + 16 * kb + 4 * g);
      },
      [&](int idx, const f32x4 (&w)[2]) {
        const int kb = idx / (HT / 2), t = 2 * (idx % (HT / 2));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          h2[t] = OSA_MFMA(w[0][s], h1[kb][s], h2[t]);
          h2[t + 1] = OSA_MFMA(w[1][s], h1[kb][s], h2[t + 1]);
        }
      });
#pragma unroll
  for (int t = 0; t < HT; ++t) h2[t] = osa_act4(h2[t], act);
#pragma unroll
  for (int o = 0; o < OT; ++o) out[o] = *reinterpret_cast<const f32x4*>(p + nl.ob3 + 16 * o + 4 * g);
  ofv_pipe<HT * OT, OFV_D>(
      [&](int idx, f32x4 (&d)[2]) {
        const int kb = idx / OT, o = idx % OT;
        d[0] = *reinterpret_cast<const f32x4*>(W3 + (16 * o + i) * nl.H + 16 * kb + 4 * g);
      },
      [&](int idx, const f32x4 (&w)[2]) {
        const int kb = idx / OT, o = idx % OT;
        out[o] = OSA_MFMA(w[0].x, h2[kb].x, out[o]);
        out[o] = OSA_MFMA(w[0].y, h2[kb].y, out[o]);
        out[o] = OSA_MFMA(w[0].z, h2[kb].z, out[o]);
        out[o] = OSA_MFMA(w[0].w, h2[kb].w, out[o]);
      });
}

template <int OT, int KBT>
__global__ __launch_bounds__(256) void osa_fvp_kernel(OsaFvpArgs a) {
  constexpr int HT = 4, NSB = 4, SPC = 64, SLD = SPC + 4, H = 64, INP = 16 * KBT, OUTP = 16 * OT;
  extern __shared__ __attribute__((aligned(16))) float ofv_smem[];
  const OsaNet& nd = a.nd;
  const OsaNet nl = ofv_padded(nd);
  float* sP = ofv_smem;           // theta, padded
  float* sV = sP + nl.P;          // v, padded
  float* sH1 = sV + nl.P;         // [H][SLD] tiles: element (feature f, sample c)
  float* sH2 = sH1 + H * SLD;
  float* sZ1 = sH2 + H * SLD;     // dL/d(pre-activation 1)
  float* sZ2 = sZ1 + H * SLD;
  float* sDO = sZ2 + H * SLD;     // [OUTP][SLD] dL/d(output)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int i = j;
  const bool vec_ok = (a.ld_obs % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.obs) & 15) == 0);
  const int P = nd.P;
  ofv_stage<KBT>(nd, nl, a.params, sP);
  ofv_stage<KBT>(nd, nl, a.vec, sV);
  __syncthreads();
  const float* __restrict__ p = sP;
  const float* __restrict__ v = sV;

  // the weight gradient of this workgroup's chunks, in registers: wave = row tile of dW1 / dW2, column tile of dW3
  f32x4 gW2[HT], gW1[KBT], gW3[OT];
#pragma unroll
  for (int t = 0; t < HT; ++t) gW2[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < KBT; ++t) gW1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < OT; ++t) gW3[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float gB = 0.f;  // thread tid < 2 H + OUTP: one bias entry
  const int nchunk = (a.M + SPC - 1) / SPC;
  bool first = true;
  // this lane's row of the chunk as S-layout fragments (zero past the end and past obs_dim); the NEXT chunk's are
  // requested as soon as the forward and tangent passes have consumed the current ones (one wave per SIMD: nothing
  // else would hide the first touch of the rows in HBM)
  // Round 6: the request is RAW -- 16-byte loads from clamped addresses, nothing computed on the values: osa_load_x
  // zeroes the columns past obs_dim with selects on the loaded data, and a select is a use: the wave sat out the whole
  // memory round trip right behind the request (1 600 of a chunk's 28 000 cycles by the phase clocks, profiles/
  // r6_fvp_phase_clocks.txt).  The masks are applied where the fragments are taken over at the top of the next chunk.
  f32x4 xn[KBT];
  bool xn_ok = false;
  auto request_rows = [&](int pos, bool have) {
    xn_ok = have && pos < a.M;
    const float* xr = xn_ok ? a.obs + (long)pos * a.ld_obs : nullptr;
    if (vec_ok) {
#pragma unroll
      for (int kb = 0; kb < KBT; ++kb) {
        const int col0 = 16 * kb + 4 * g;
        const bool ok = xn_ok && col0 < nd.obs_dim;  // (ld % 4 == 0 and col0 < obs_dim <= ld: the piece lies inside the row)
        xn[kb] = *reinterpret_cast<const f32x4*>(ok ? xr + col0 : a.obs);
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < KBT; ++kb) xn[kb] = osa_load_x(xr, 16 * kb + 4 * g, nd.obs_dim, a.ld_obs, false);
    }
  };
  auto take_rows = [&](f32x4 (&xf)[KBT]) {  // osa_load_x's values
#pragma unroll
    for (int kb = 0; kb < KBT; ++kb) {
      f32x4 x = xn[kb];
      if (vec_ok) {
        const int col0 = 16 * kb + 4 * g;
        if (!(xn_ok && col0 < nd.obs_dim)) x.x = 0.f;
        if (!(xn_ok && col0 + 1 < nd.obs_dim)) x.y = 0.f;
        if (!(xn_ok && col0 + 2 < nd.obs_dim)) x.z = 0.f;
        if (!(xn_ok && col0 + 3 < nd.obs_dim)) x.w = 0.f;
      }
      xf[kb] = x;
    }
  };
  request_rows(blockIdx.x * SPC + 16 * wave + j, true);
#ifdef OFV_CLOCKS
  long long ofv_acc_[12] = {};
  long long ofv_t_ = (long long)__builtin_amdgcn_s_memtime();
#endif
  for (int chunk = blockIdx.x; chunk < nchunk; chunk += a.nblk, first = false) {
    OFV_MARK(0);  // (loop overhead, the previous chunk's tail)
    const int pos = chunk * SPC + 16 * wave + j;
    const bool valid = pos < a.M;
    f32x4 xf[KBT];
    take_rows(xf);
    __syncthreads();  // previous chunk's tiles fully consumed
    OFV_MARK(1);

    f32x4 h1[HT], h2[HT], out[OT];
    ofv_forward<OT, KBT>(nl, p, xf, OSA_ACT_TANH, h1, h2, out);  // (tanh only: a run-time activation switch is a branch,
    // i.e. a scheduling barrier, between the MFMA groups of every layer; other activations keep the general kernel)
    OFV_MARK(2);
    // ---- JVP: t = d(mean) along v
    f32x4 dO[OT];
#pragma unroll
    for (int o = 0; o < OT; ++o) dO[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      f32x4 t1[HT], t2[HT], tm[OT];
#pragma unroll
      for (int t = 0; t < HT; ++t) t1[t] = *reinterpret_cast<const f32x4*>(v + nl.ob1 + 16 * t + 4 * g);
      ofv_pipe<KBT * HT / 2, OFV_D>(
          [&](int idx, f32x4 (&d)[2]) {
            const int kb = idx / (HT / 2), t = 2 * (idx % (HT / 2));
            d[0] = *reinterpret_cast<const f32x4*>(v + nl.oW1 + (16 * t + i) * nl.INP + 16 * kb + 4 * g);
            d[1] = *reinterpret_cast<const f32x4*>(v + nl.oW1 + (16 * (t + 1) + i) * nl.INP + 16 * kb + 4 * g);
          },
          [&](int idx, const f32x4 (&w)[2]) {
            const int kb = idx / (HT / 2), t = 2 * (idx % (HT / 2));
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              t1[t] = OSA_MFMA(w[0][s], xf[kb][s], t1[t]);
              t1[t + 1] = OSA_MFMA(w[1][s], xf[kb][s], t1[t + 1]);
            }
          });
#pragma unroll
      for (int t = 0; t < HT; ++t) t1[t] = t1[t] * osa_dact4(h1[t], OSA_ACT_TANH);
#pragma unroll
      for (int t = 0; t < HT; ++t) t2[t] = *reinterpret_cast<const f32x4*>(v + nl.ob2 + 16 * t + 4 * g);
      ofv_pipe<HT * HT, OFV_D>(
          [&](int idx, f32x4 (&d)[2]) {
            const int kb = idx / HT, t = idx % HT;
            d[0] = *reinterpret_cast<const f32x4*>(v + nl.oW2 + (16 * t + i) * nl.H + 16 * kb + 4 * g);
            d[1] = *reinterpret_cast<const f32x4*>(p + nl.oW2 + (16 * t + i) * nl.H + 16 * kb + 4 * g);
          },
          [&](int idx, const f32x4 (&w)[2]) {
            const int kb = idx / HT, t = idx % HT;
            t2[t] = OSA_MFMA(w[0].x, h1[kb].x, t2[t]);
            t2[t] = OSA_MFMA(w[0].y, h1[kb].y, t2[t]);
            t2[t] = OSA_MFMA(w[0].z, h1[kb].z, t2[t]);
            t2[t] = OSA_MFMA(w[0].w, h1[kb].w, t2[t]);
            t2[t] = OSA_MFMA(w[1].x, t1[kb].x, t2[t]);
            t2[t] = OSA_MFMA(w[1].y, t1[kb].y, t2[t]);
            t2[t] = OSA_MFMA(w[1].z, t1[kb].z, t2[t]);
            t2[t] = OSA_MFMA(w[1].w, t1[kb].w, t2[t]);
          });
#pragma unroll
      for (int t = 0; t < HT; ++t) t2[t] = t2[t] * osa_dact4(h2[t], OSA_ACT_TANH);
#pragma unroll
      for (int o = 0; o < OT; ++o) tm[o] = *reinterpret_cast<const f32x4*>(v + nl.ob3 + 16 * o + 4 * g);
      ofv_pipe<HT * OT, OFV_D>(
          [&](int idx, f32x4 (&d)[2]) {
            const int kb = idx / OT, o = idx % OT;
            d[0] = *reinterpret_cast<const f32x4*>(v + nl.oW3 + (16 * o + i) * nl.H + 16 * kb + 4 * g);
            d[1] = *reinterpret_cast<const f32x4*>(p + nl.oW3 + (16 * o + i) * nl.H + 16 * kb + 4 * g);
          },
          [&](int idx, const f32x4 (&w)[2]) {
            const int kb = idx / OT, o = idx % OT;
            tm[o] = OSA_MFMA(w[0].x, h2[kb].x, tm[o]);
            tm[o] = OSA_MFMA(w[0].y, h2[kb].y, tm[o]);
            tm[o] = OSA_MFMA(w[0].z, h2[kb].z, tm[o]);
            tm[o] = OSA_MFMA(w[0].w, h2[kb].w, tm[o]);
            tm[o] = OSA_MFMA(w[1].x, t2[kb].x, tm[o]);
            tm[o] = OSA_MFMA(w[1].y, t2[kb].y, tm[o]);
            tm[o] = OSA_MFMA(w[1].z, t2[kb].z, tm[o]);
            tm[o] = OSA_MFMA(w[1].w, t2[kb].w, tm[o]);
          });
#pragma unroll
      for (int o = 0; o < OT; ++o) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = 16 * o + 4 * g + r;
          if (d < nd.act_dim && valid) {
            const float sd = expf(p[nl.oLS + d]);
            dO[o][r] = tm[o][r] / (sd * sd) * a.fvp_scale;
          }
        }
      }
    }
    OFV_MARK(3);
    {  // the next chunk's rows (the loads ride under the backward pass and the contractions)
      request_rows((chunk + a.nblk) * SPC + 16 * wave + j, chunk + a.nblk < nchunk);
    }
    OFV_MARK(4);
    // ---- backward through the hidden layers (S layout, activations stay in registers)
    const float* __restrict__ W2 = p + nl.oW2;
    const float* __restrict__ W3 = p + nl.oW3;
    f32x4 z2[HT], z1[HT];
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      ofv_pipe<HT * OT, OFV_D>(
          [&](int idx, f32x4 (&d)[2]) {  // A[i][k] = W3^T[16t+i][16o+4g+s]
            const int t = idx / OT, o = idx % OT;
#pragma unroll
            for (int s = 0; s < 4; ++s) d[0][s] = W3[(16 * o + 4 * g + s) * nl.H + 16 * t + i];
          },
          [&](int idx, const f32x4 (&w)[2]) {
            const int t = idx / OT, o = idx % OT;
            if (o == 0) acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = OSA_MFMA(w[0][s], dO[o][s], acc);
            if (o == OT - 1) z2[t] = acc * osa_dact4(h2[t], OSA_ACT_TANH);
          });
    }
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      ofv_pipe<HT * HT, OFV_D>(
          [&](int idx, f32x4 (&d)[2]) {  // A[i][k] = W2^T[16t+i][16kb+4g+s]
            const int t = idx / HT, kb = idx % HT;
#pragma unroll
            for (int s = 0; s < 4; ++s) d[0][s] = W2[(16 * kb + 4 * g + s) * nl.H + 16 * t + i];
          },
          [&](int idx, const f32x4 (&w)[2]) {
            const int t = idx / HT, kb = idx % HT;
            if (kb == 0) acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = OSA_MFMA(w[0][s], z2[kb][s], acc);
            if (kb == HT - 1) z1[t] = acc * osa_dact4(h1[t], OSA_ACT_TANH);
          });
    }
    OFV_MARK(5);
    // ---- S layout -> F layout through LDS: element (feature f, sample c) at [f * SLD + c]
    const int c = 16 * wave + j;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 16 * t + 4 * g + r;
        sH1[f * SLD + c] = h1[t][r];
        sH2[f * SLD + c] = h2[t][r];
        sZ1[f * SLD + c] = z1[t][r];
        sZ2[f * SLD + c] = z2[t][r];
      }
    }
#pragma unroll
    for (int o = 0; o < OT; ++o) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sDO[(16 * o + 4 * g + r) * SLD + c] = dO[o][r];
    }
    __syncthreads();
    OFV_MARK(6);
    // ---- weight gradients: contraction over the 64 samples of the chunk.  D tile: lane (cc = l & 15, g) holds
    // dW[row 4g + r][col cc]; wave w owns row tile w of dW2 and dW1 (all column tiles) and column tile w of dW3
    f32x4 a1[NSB];
    {
      const int rt = wave;
      f32x4 a2[NSB];
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb) {
        a2[sb] = *reinterpret_cast<const f32x4*>(sZ2 + (16 * rt + i) * SLD + 16 * sb + 4 * g);
        a1[sb] = *reinterpret_cast<const f32x4*>(sZ1 + (16 * rt + i) * SLD + 16 * sb + 4 * g);
      }
      {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        ofv_pipe<HT * NSB, OFV_D>(
            [&](int idx, f32x4 (&d)[2]) {
              const int ti = idx / NSB, sb = idx % NSB;
              d[0] = *reinterpret_cast<const f32x4*>(sH1 + (16 * ti + i) * SLD + 16 * sb + 4 * g);
            },
            [&](int idx, const f32x4 (&b)[2]) {
              const int ti = idx / NSB, sb = idx % NSB;
              if (sb == 0) acc = (f32x4){0.f, 0.f, 0.f, 0.f};
              acc = OSA_MFMA(a2[sb].x, b[0].x, acc);
              acc = OSA_MFMA(a2[sb].y, b[0].y, acc);
              acc = OSA_MFMA(a2[sb].z, b[0].z, acc);
              acc = OSA_MFMA(a2[sb].w, b[0].w, acc);
              if (sb == NSB - 1) gW2[ti] = first ? acc : gW2[ti] + acc;
            });
      }
    }
    OFV_MARK(7);
    {  // dW3: output tiles o x column tile `wave`
      const int ct = wave;
      {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        ofv_pipe<OT * NSB, OFV_D>(
            [&](int idx, f32x4 (&d)[2]) {
              const int o = idx / NSB, sb = idx % NSB;
              d[0] = *reinterpret_cast<const f32x4*>(sDO + (16 * o + i) * SLD + 16 * sb + 4 * g);
              d[1] = *reinterpret_cast<const f32x4*>(sH2 + (16 * ct + i) * SLD + 16 * sb + 4 * g);
            },
            [&](int idx, const f32x4 (&q)[2]) {
              const int o = idx / NSB, sb = idx % NSB;
              if (sb == 0) acc = (f32x4){0.f, 0.f, 0.f, 0.f};
              acc = OSA_MFMA(q[0].x, q[1].x, acc);
              acc = OSA_MFMA(q[0].y, q[1].y, acc);
              acc = OSA_MFMA(q[0].z, q[1].z, acc);
              acc = OSA_MFMA(q[0].w, q[1].w, acc);
              if (sb == NSB - 1) gW3[o] = first ? acc : gW3[o] + acc;
            });
      }
    }
    OFV_MARK(8);
    // bias gradients: one thread per feature sums its LDS row over the 64 samples
    if ((int)threadIdx.x < 2 * H + OUTP) {
      const int tid = threadIdx.x;
      const float* srow = tid < H ? sZ1 + tid * SLD : (tid < 2 * H ? sZ2 + (tid - H) * SLD : sDO + (tid - 2 * H) * SLD);
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < SPC; k += 4) {  // (16-byte reads; the additions in sample order, as the general kernel's)
        const f32x4 q = *reinterpret_cast<const f32x4*>(srow + k);
        s += q.x;
        s += q.y;
        s += q.z;
        s += q.w;
      }
      gB = first ? s : gB + s;
    }
    OFV_MARK(9);
    // dW1's second operand -- the chunk's rows, element (input feature f, sample c) -- takes the place of the h1 / h2
    // tiles once dW2 and dW3 have consumed them (the general kernel gathers these values from global memory again, 64 scalar loads per
    // lane; the 155 KB of parameters and tiles leave no room for a fifth tile): same values, same MFMA order
    __syncthreads();
    float* sX = sH1;  // (the h1 and h2 tiles are adjacent and both consumed: up to 128 input features)
#pragma unroll
    for (int kb = 0; kb < KBT; ++kb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sX[(16 * kb + 4 * g + r) * SLD + c] = xf[kb][r];
    }
    __syncthreads();
    OFV_MARK(10);
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      ofv_pipe<KBT * NSB, OFV_D>(
          [&](int idx, f32x4 (&d)[2]) {  // B[k = sample 16 sb + 4 g + s][j = input feature 16 kb + cc]
            const int kb = idx / NSB, sb = idx % NSB;
            d[0] = *reinterpret_cast<const f32x4*>(sX + (16 * kb + i) * SLD + 16 * sb + 4 * g);
          },
          [&](int idx, const f32x4 (&b)[2]) {
            const int kb = idx / NSB, sb = idx % NSB;
            if (sb == 0) acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc = OSA_MFMA(a1[sb].x, b[0].x, acc);
            acc = OSA_MFMA(a1[sb].y, b[0].y, acc);
            acc = OSA_MFMA(a1[sb].z, b[0].z, acc);
            acc = OSA_MFMA(a1[sb].w, b[0].w, acc);
            if (sb == NSB - 1) gW1[kb] = first ? acc : gW1[kb] + acc;
          });
    }
    OFV_MARK(11);
  }  // chunks
#ifdef OFV_CLOCKS
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int k = 0; k < 12; ++k) ofv_clk[k] += ofv_acc_[k];
#endif
  // ---- this workgroup's slab
  float* __restrict__ gout = a.slabs + (long)blockIdx.x * (P + OFV_NSTAT);
  const int cc = j;
#pragma unroll
  for (int ti = 0; ti < HT; ++ti) {
#pragma unroll
    for (int r = 0; r < 4; ++r) gout[nd.oW2 + (16 * wave + 4 * g + r) * H + 16 * ti + cc] = gW2[ti][r];
  }
#pragma unroll
  for (int kb = 0; kb < KBT; ++kb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) gout[nd.oW1 + (16 * wave + 4 * g + r) * INP + 16 * kb + cc] = gW1[kb][r];
  }
#pragma unroll
  for (int o = 0; o < OT; ++o) {
#pragma unroll
    for (int r = 0; r < 4; ++r) gout[nd.oW3 + (16 * o + 4 * g + r) * H + 16 * wave + cc] = gW3[o][r];
  }
  {
    const int tid = threadIdx.x;
    if (tid < H) gout[nd.ob1 + tid] = gB;
    else if (tid < 2 * H) gout[nd.ob2 + (tid - H)] = gB;
    else if (tid < 2 * H + OUTP) gout[nd.ob3 + (tid - 2 * H)] = gB;
    else if (tid < 2 * H + 2 * OUTP) gout[nd.oLS + (tid - 2 * H - OUTP)] = 0.f;  // (log_std: osa_fvp_finish adds 2 / D_a)
  }
}

// slab b goes to partial b mod 8, the eight partials combined in a fixed order: osa_slab_reduce_kernel's sum
__global__ __launch_bounds__(256) void osa_fvp_reduce_kernel(OsaFvpArgs a) {
  const OsaNet& nd = a.nd;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int W = nd.P + OFV_NSTAT;
  if (e >= nd.P) {
    if (e == nd.P) {  // the statistics slots osa_slab_reduce_kernel fills for the actor (a product has no loss)
      float ent = 0.f;
      for (int d = 0; d < nd.act_dim; ++d) ent += 1.41893853320467274178f + a.params[nd.oLS + d];
      ent /= (float)nd.act_dim;
      a.stats[2] = 0.f;
      a.stats[3] = 0.f;
      a.stats[4] = ent;
    }
    return;
  }
  const float* s = a.slabs + e;
  const int ns = a.nblk;
  float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int b = 0;
  for (; b + 64 <= ns; b += 64) {  // 64 slabs' loads in flight (the loop is nothing but latency: 256 slabs = 4 round trips
    // instead of 32); the additions keep the order of the 8-slab loop: partial u receives b + u, b + 8 + u, ...
    float t[64];
#pragma unroll
    for (int u = 0; u < 64; ++u) t[u] = s[(long)(b + u) * W];
#pragma unroll
    for (int v = 0; v < 64; v += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] += t[v + u];
    }
  }
  for (; b + 8 <= ns; b += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] += s[(long)(b + u) * W];
  }
  for (; b < ns; ++b) p[0] += s[(long)b * W];
  a.grads[e] = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
}

// launcher for osa_actor_fvp_raw (mlp_kernels.hip): OSA_EUNSUPPORTED = shape outside this kernel, use the general one
int osa_launch_fvp_fast(const OsaNet& nd, const float* params, float* grads, const float* obs, int ld_obs, long M,
                        const float* vec, int max_blocks, float* ws, float* step_stats, hipStream_t st) {
  const char* sw = getenv("OSA_FVP_FAST");  // (read per call: the parity test switches between the two kernels)
  const bool off = sw != nullptr && sw[0] == '0' && sw[1] == 0;
  if (off) return OSA_EUNSUPPORTED;
  if (nd.act != OSA_ACT_TANH || nd.H != 64 || nd.KB > 5 || nd.OUTP > 32 || M <= 64 || M > (1l << 30)) return OSA_EUNSUPPORTED;
  OsaFvpArgs a = {};
  a.nd = nd; a.params = params; a.vec = vec; a.obs = obs; a.ld_obs = ld_obs; a.M = (int)M;
  a.grads = grads; a.stats = step_stats; a.slabs = ws;
  a.fvp_scale = (float)(1.0 / ((double)M * nd.act_dim));
  const int nchunk = (int)((M + 63) / 64);
  int nblk = nchunk;
  if (max_blocks < 1) max_blocks = 1;
  if (nblk > max_blocks) nblk = max_blocks;
  a.nblk = nblk;
  const int H = 64, SLD = 68;
  const size_t padded = (size_t)H * (nd.INP + 4) + H + (size_t)H * (H + 4) + H + (size_t)nd.OUTP * (H + 4) + 2 * nd.OUTP;
  const size_t lds = (2 * padded + (size_t)4 * H * SLD + (size_t)nd.OUTP * SLD) * sizeof(float);
  if (lds > 160 * 1024) return OSA_EUNSUPPORTED;
#define OFV_GO(OT, KBT)                                                                                          \
  do {                                                                                                          \
    static OsaPerDeviceOnce attr_set;                                                                               \
    if (attr_set.need()) {                                                                                            \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&osa_fvp_kernel<OT, KBT>),                          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)            \
        return OSA_EHIP;                                                                                        \
      attr_set.set();                                                                                          \
    }                                                                                                           \
    hipLaunchKernelGGL((osa_fvp_kernel<OT, KBT>), dim3(nblk), dim3(256), lds, st, a);                           \
  } while (0)
  const int OT = nd.OUTP / 16;
  if (OT == 1) {
    if (nd.KB == 1) OFV_GO(1, 1); else if (nd.KB == 2) OFV_GO(1, 2); else if (nd.KB == 3) OFV_GO(1, 3);
    else if (nd.KB == 4) OFV_GO(1, 4); else OFV_GO(1, 5);
  } else {
    if (nd.KB == 1) OFV_GO(2, 1); else if (nd.KB == 2) OFV_GO(2, 2); else if (nd.KB == 3) OFV_GO(2, 3);
    else if (nd.KB == 4) OFV_GO(2, 4); else OFV_GO(2, 5);
  }
#undef OFV_GO
  hipLaunchKernelGGL(osa_fvp_reduce_kernel, dim3((nd.P + 1 + 255) / 256), dim3(256), 0, st, a);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}
