// Rollout-buffer kernels: store (K4), dual reward+cost GAE backward scan (K5), advantage statistics
// and env-major get() (K6).  HBM-bound byte work: coalesced along the env axis n of the time-major
// (T, N) layout, float64 recurrences exactly as the reference (see include/omnisafe_amd.h).
#include "osa_common.h"

// ------------------------------------------------------------------------------------------------
// K4  store one vector step into row t
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void osa_store_step_kernel(
    int N, int obs_dim, int act_dim, const float* __restrict__ obs, int ld_os,
    const float* __restrict__ act, int ld_as, const float* __restrict__ reward,
    const float* __restrict__ cost, const float* __restrict__ value_r,
    const float* __restrict__ value_c, const float* __restrict__ logp, float* __restrict__ b_obs,
    int ld_ob, float* __restrict__ b_act, int ld_ab, float* __restrict__ b_reward,
    float* __restrict__ b_cost, float* __restrict__ b_value_r, float* __restrict__ b_value_c,
    float* __restrict__ b_logp) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n_obs = (long)N * obs_dim, n_act = (long)N * act_dim;
  if (gid < n_obs) {
    const int n = (int)(gid / obs_dim), d = (int)(gid % obs_dim);
    b_obs[(long)n * ld_ob + d] = obs[(long)n * ld_os + d];
  }
  if (gid < n_act) {
    const int n = (int)(gid / act_dim), d = (int)(gid % act_dim);
    b_act[(long)n * ld_ab + d] = act[(long)n * ld_as + d];
  }
  if (gid < N) {
    b_reward[gid] = reward[gid];
    b_cost[gid] = cost[gid];
    b_value_r[gid] = value_r[gid];
    b_value_c[gid] = value_c[gid];
    b_logp[gid] = logp[gid];
  }
}

// ------------------------------------------------------------------------------------------------
// K5  dual GAE backward scan.  One lane per env; each wave reads 64 consecutive envs of row t
// (256 B coalesced per array).  The time axis is walked in chunks of U steps whose loads are all
// issued before the (sequential, float64) recurrence consumes them, so HBM latency is paid once per
// chunk instead of once per step.
// ------------------------------------------------------------------------------------------------
template <int EST, int U>
__global__ __launch_bounds__(256) void osa_gae_scan_kernel(
    const float* __restrict__ reward, const float* __restrict__ cost,
    const float* __restrict__ value_r, const float* __restrict__ value_c,
    const uint8_t* __restrict__ path_end, const float* __restrict__ boot_r,
    const float* __restrict__ boot_c, int T, int N, float g32, double d_g, double d_r, double d_c,
    float pc, float* __restrict__ adv_r, float* __restrict__ adv_c, float* __restrict__ tgt_r,
    float* __restrict__ tgt_c, float* __restrict__ disc_ret) {
#pragma clang fp contract(off)
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float nv_r = 0.f, nv_c = 0.f;
  float vs_r = 0.f, vs_c = 0.f;  // v-trace: v_s of the following step (float32 recursion)
  double a_r = 0.0, a_c = 0.0, ret = 0.0, rtg_r = 0.0, rtg_c = 0.0;
  for (int t0 = T - 1; t0 >= 0; t0 -= U) {
    float r[U], c[U], vr[U], vc[U], br[U], bc[U];
    uint8_t e[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 - u;
      if (t >= 0) {
        const long i = (long)t * N + n;
        r[u] = reward[i];
        c[u] = cost[i];
        vr[u] = value_r[i];
        vc[u] = value_c[i];
        e[u] = path_end[i];
        br[u] = e[u] ? boot_r[i] : 0.f;
        bc[u] = e[u] ? boot_c[i] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 - u;
      if (t < 0) break;
      if (e[u]) {  // a path ends after step t: re-seed every carry with the bootstrap
        nv_r = br[u];
        nv_c = bc[u];
        a_r = 0.0;
        a_c = 0.0;
        ret = (double)br[u];
        const float pb = pc * bc[u];
        rtg_r = (double)(br[u] - pb);
        rtg_c = (double)bc[u];
        vs_r = br[u];  // v-trace: last_v_s = values[-1] (un-penalised bootstrap, onpolicy_buffer.py:397)
        vs_c = bc[u];
      }
      const float pcost = pc * c[u];
      const float r_pen = r[u] - pcost;
      const float gr = g32 * nv_r;
      const float gc = g32 * nv_c;
      const float sr = r_pen + gr;
      const float sc = c[u] + gc;
      const float delta_r = sr - vr[u];
      const float delta_c = sc - vc[u];
      const double m0 = d_g * ret;
      ret = (double)r[u] + m0;
      double out_ar, out_ac, out_tr, out_tc;
      if (EST == OSA_EST_VTRACE) {
        // OnPolicyBuffer._calculate_v_trace (onpolicy_buffer.py:380-405) with the reference's call
        // arguments: policy == behaviour probabilities, so rho = c = 1 (a float32 recursion):
        //   delta = r + g*v' - v;  v_s = v + (delta + g*(v_s' - v'));  adv = (r + g*v_s') - v
        // (v' and v_s' are both seeded with values[-1] = the bootstrap at a path end, see above;
        // rewards[-1], the penalised bootstrap, is never read by the recursion)
        const float adv_vr = (r_pen + g32 * vs_r) - vr[u];
        const float adv_vc = (c[u] + g32 * vs_c) - vc[u];
        const float t3r = vs_r - nv_r, t3c = vs_c - nv_c;
        const float t4r = g32 * t3r, t4c = g32 * t3c;
        const float t5r = delta_r + t4r, t5c = delta_c + t4c;
        vs_r = vr[u] + t5r;
        vs_c = vc[u] + t5c;
        const long iv = (long)t * N + n;
        adv_r[iv] = adv_vr;
        adv_c[iv] = adv_vc;
        tgt_r[iv] = vs_r;
        tgt_c[iv] = vs_c;
        disc_ret[iv] = (float)ret;
        nv_r = vr[u];
        nv_c = vc[u];
        continue;
      }
      if (EST != OSA_EST_GAE) {
        const double m1 = d_g * rtg_r;
        rtg_r = (double)r_pen + m1;
        const double m2 = d_g * rtg_c;
        rtg_c = (double)c[u] + m2;
      }
      if (EST == OSA_EST_PLAIN) {
        out_ar = (double)delta_r;
        out_ac = (double)delta_c;
      } else {
        const double m3 = d_r * a_r;
        a_r = (double)delta_r + m3;
        const double m4 = d_c * a_c;
        a_c = (double)delta_c + m4;
        out_ar = a_r;
        out_ac = a_c;
      }
      if (EST == OSA_EST_GAE) {
        out_tr = out_ar + (double)vr[u];
        out_tc = out_ac + (double)vc[u];
      } else {
        out_tr = rtg_r;
        out_tc = rtg_c;
      }
      const long i = (long)t * N + n;
      adv_r[i] = (float)out_ar;
      adv_c[i] = (float)out_ac;
      tgt_r[i] = (float)out_tr;
      tgt_c[i] = (float)out_tc;
      disc_ret[i] = (float)ret;
      nv_r = vr[u];
      nv_c = vc[u];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K5b  the same dual GAE scan, parallel over TIME: the wavefront backward scan with LDS staging.
//
// The lane-per-env kernel above needs N >= ~32 k envs to fill the chip and walks T sequentially; for few
// envs / long horizons (BASELINE config 1: N = 4, T = 5000) almost every lane of the GPU idles.  Here a
// workgroup owns NB = 16 consecutive envs and walks the time axis backwards in tiles of 64 steps:
//   1. the tile (64 steps x 16 envs of r, c, v_r, v_c, path_end, boot_r, boot_c) is fetched with coalesced
//      row segments (16 envs x 4 B = 64 B per step and array; the next tile's loads are issued into registers
//      before the current one is processed) and staged in LDS;
//   2. each wave takes 4 of the envs; for one env the 64 lanes ARE the 64 time steps (lane 0 = latest).
//      Every recurrence y_t = x_t + c_t * y_{t+1} (c_t = gamma*lambda, or 0 where a path ends) is an affine
//      map, so the strip is a scan of (c, x) pairs in 6 DPP rounds (row_shr 1/2/4/8, row_bcast 15/31); because
//      c_t is either 0 or one constant d, the composed coefficient of a window of k steps is d^k or 0,
//      decided per lane from the wave's ballot of path ends -- only the x parts travel between lanes.  The
//      carry of the tile (lane 63) continues into the next (earlier) tile;
//   3. results go back through LDS and leave with the same coalesced row segments.
// Arithmetic: float32 deltas exactly as the lane-per-env kernel; the recurrences in float64 but associated
// as a tree instead of a chain, so results agree with the bit-exact kernel to float64 rounding (then rounded
// to float32: identical in all but ~1e-7 of the elements; tests require rtol 1e-5).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double osa_readlane63_f64(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
// DPP move of a double (two dwords); lanes without a valid source receive +0.0
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double osa_dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Per-lane powers of one recurrence coefficient d, computed once per kernel.
struct OsaScanPow {
  double o1, o2, o4, o8;  // d^1, d^2, d^4, d^8 (wave-uniform)
  double row;             // d^((lane & 15) + 1): joins the prefix that ends right before the lane's row of 16
  double half;            // d^(lane - 31) for lanes >= 32: joins the prefix of lanes 0..31
  double all;             // d^(lane + 1): joins the carry of the previous tile
};
__device__ __forceinline__ double osa_powi(double d, int e) {
  double r = 1.0, b = d;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    if ((e >> k) & 1) r *= b;
    b *= b;
  }
  return r;
}
__device__ __forceinline__ OsaScanPow osa_scan_pow(double d, int lane) {
  OsaScanPow p;
  p.o1 = d; p.o2 = d * d; p.o4 = p.o2 * p.o2; p.o8 = p.o4 * p.o4;
  p.row = osa_powi(d, (lane & 15) + 1);
  p.half = osa_powi(d, lane >= 32 ? lane - 31 : 0);
  p.all = osa_powi(d, lane + 1);
  return p;
}

// Inclusive scan of y_l = x_l + (reset_l ? 0 : d) * y_{l-1} over the 64 lanes of a wave (lane 0 first),
// y_{-1} = carry.  `resets` = ballot of reset_l.  Because the coefficient of every step is either d or 0,
// the composed coefficient of a window of k steps is d^k if the window holds no reset and 0 otherwise:
// only the x parts move between lanes -- on the DPP network: row_shr 1, 2, 4, 8 inside each row of 16
// lanes, then row_bcast:15 and row_bcast:31 across rows (the gfx9 wave64 scan sequence).
__device__ __forceinline__ double osa_affine_scan64(double x, unsigned long long resets, const OsaScanPow& p,
                                                    double carry, int lane) {
#pragma clang fp contract(off)
  const int lr = lane & 15;
#define OSA_SCAN_ROW_STEP(O, CTRL, PW)                                                              \
  {                                                                                                 \
    const double up = osa_dpp_f64<CTRL, 0xf>(x);                                                    \
    /* window of this lane before the step: lanes [lane-O+1 .. lane], all inside its row if lr >= O */ \
    const bool open = lr >= (O) && ((resets >> ((lane - (O) + 1) & 63)) & ((1ull << (O)) - 1ull)) == 0ull; \
    const double m = (PW) * up;  /* (a select below, not a product with 0: NaNs stop at path ends) */ \
    x = open ? x + m : x;                                                                           \
  }
  OSA_SCAN_ROW_STEP(1, 0x111, p.o1)
  OSA_SCAN_ROW_STEP(2, 0x112, p.o2)
  OSA_SCAN_ROW_STEP(4, 0x114, p.o4)
  OSA_SCAN_ROW_STEP(8, 0x118, p.o8)
#undef OSA_SCAN_ROW_STEP
  {  // rows 1 and 3 take the total of the row before them (its lane 15)
    const double up = osa_dpp_f64<0x142, 0xa>(x);
    const int rs = lane & ~15;
    const bool open = ((lane >> 4) & 1) && ((resets >> rs) & ((2ull << lr) - 1ull)) == 0ull;
    const double m = p.row * up;
    x = open ? x + m : x;
  }
  {  // rows 2 and 3 take the total of lanes 0..31 (lane 31)
    const double up = osa_dpp_f64<0x143, 0xc>(x);
    const bool open = lane >= 32 && ((resets >> 32) & ((2ull << (lane - 32)) - 1ull)) == 0ull;
    const double m = p.half * up;
    x = open ? x + m : x;
  }
  {
    const unsigned long long all = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    const double m = p.all * carry;
    x = (resets & all) == 0ull ? x + m : x;
  }
  return x;
}

// NWV waves per workgroup: 4 (one per SIMD) when the grid alone fills the chip; 16 (four per SIMD, ONE env per
// wave) when N / NB workgroups are at most ~ 2 per compute unit -- the scan of a tile is a chain of dependent
// float64 DPP rounds, and with one wave per SIMD nothing hides their latency (measured at T = 4096, N = 4096:
// 6.5 us per 36 KB tile and workgroup, i.e. compute-latency-bound at 1.4 TB/s)
template <int EST, int NB, int NWV>
__global__ __launch_bounds__(64 * NWV) void osa_gae_tile_scan_kernel(
    const float* __restrict__ reward, const float* __restrict__ cost,
    const float* __restrict__ value_r, const float* __restrict__ value_c,
    const uint8_t* __restrict__ path_end, const float* __restrict__ boot_r,
    const float* __restrict__ boot_c, int T, int N, float g32, double d_g, double d_r, double d_c,
    float pc, float* __restrict__ adv_r, float* __restrict__ adv_c, float* __restrict__ tgt_r,
    float* __restrict__ tgt_c, float* __restrict__ disc_ret) {
#pragma clang fp contract(off)
  constexpr int TT = 64, LD = NB + 1, NTH = 64 * NWV, EPW = NB / NWV, PER = TT * NB / NTH;  // envs per wave, elements per thread
  __shared__ float s_in[7][TT][LD];   // r, c, v_r, v_c, boot_r, boot_c, path_end (as 0/1)
  __shared__ float s_out[5][TT][LD];  // adv_r, adv_c, tgt_r, tgt_c, disc_ret
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * NB;
  const OsaScanPow pw_r = osa_scan_pow(d_r, lane), pw_c = osa_scan_pow(d_c, lane), pw_g = osa_scan_pow(d_g, lane);
  // per-env carries of this wave (wave-uniform values)
  double ca_r[EPW], ca_c[EPW], cret[EPW], crtg_r[EPW], crtg_c[EPW];
  float cnv_r[EPW], cnv_c[EPW];
#pragma unroll
  for (int q = 0; q < EPW; ++q) {
    ca_r[q] = ca_c[q] = cret[q] = crtg_r[q] = crtg_c[q] = 0.0;
    cnv_r[q] = cnv_c[q] = 0.f;
  }
  const int ntiles = (T + TT - 1) / TT;
  float pre[7][PER];
  auto fetch = [&](int j) {  // tile j: rows tt = 0.. <-> t = t_hi - tt, t_hi = T-1-64j
    const int t_hi = T - 1 - j * TT;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + NTH * k, tt = i / NB, e = i % NB;
      const int t = t_hi - tt, n = n0 + e;
      const bool ok = t >= 0 && n < N;
      const long g = ok ? (long)t * N + n : 0;
      pre[0][k] = ok ? reward[g] : 0.f;
      pre[1][k] = ok ? cost[g] : 0.f;
      pre[2][k] = ok ? value_r[g] : 0.f;
      pre[3][k] = ok ? value_c[g] : 0.f;
      const bool pe = ok && path_end[g] != 0;
      pre[4][k] = pe ? boot_r[g] : 0.f;
      pre[5][k] = pe ? boot_c[g] : 0.f;
      pre[6][k] = pe ? 1.f : 0.f;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + NTH * k, tt = i / NB, e = i % NB;
#pragma unroll
      for (int a = 0; a < 7; ++a) s_in[a][tt][e] = pre[a][k];
    }
  };
  fetch(0);
  stage();
  __syncthreads();
  for (int j = 0; j < ntiles; ++j) {
    const int t_hi = T - 1 - j * TT;
    const int nvalid = t_hi + 1 < TT ? t_hi + 1 : TT;
    if (j + 1 < ntiles) fetch(j + 1);  // in flight while this tile is scanned
#pragma unroll
    for (int q = 0; q < EPW; ++q) {
      const int e = q * NWV + wave;  // envs interleaved over the waves: few envs still use all of them
      if (n0 + e >= N) continue;   // wave-uniform
      const float r = s_in[0][lane][e], c = s_in[1][lane][e];
      const float vr = s_in[2][lane][e], vc = s_in[3][lane][e];
      const float br = s_in[4][lane][e], bc = s_in[5][lane][e];
      // a path ends after step t; the last row of the buffer without a flag = bootstrap 0 (as the
      // lane-per-env kernel, whose carries start at 0)
      const bool pe = s_in[6][lane][e] != 0.f || (j == 0 && lane == 0);
      const unsigned long long resets = __ballot(pe);
      const float nxt_r = lane == 0 ? cnv_r[q] : s_in[2][(lane + 63) & 63][e];
      const float nxt_c = lane == 0 ? cnv_c[q] : s_in[3][(lane + 63) & 63][e];
      const float nv_r = pe ? br : nxt_r, nv_c = pe ? bc : nxt_c;
      const float pcost = pc * c;
      const float r_pen = r - pcost;
      const float gr = g32 * nv_r, gc = g32 * nv_c;
      const float sr = r_pen + gr, sc = c + gc;
      const float delta_r = sr - vr, delta_c = sc - vc;
      // discounted return: x = r (+ gamma * bootstrap where a path ends)
      double x_ret = (double)r;
      if (pe) { const double m = d_g * (double)br; x_ret = x_ret + m; }
      const double ret = osa_affine_scan64(x_ret, resets, pw_g, cret[q], lane);
      double o_ar, o_ac, o_tr, o_tc;
      if (EST == OSA_EST_GAE || EST == OSA_EST_GAE_RTG) {
        o_ar = osa_affine_scan64((double)delta_r, resets, pw_r, ca_r[q], lane);
        o_ac = osa_affine_scan64((double)delta_c, resets, pw_c, ca_c[q], lane);
        ca_r[q] = osa_readlane63_f64(o_ar);
        ca_c[q] = osa_readlane63_f64(o_ac);
      } else {
        o_ar = (double)delta_r;
        o_ac = (double)delta_c;
      }
      if (EST == OSA_EST_GAE) {
        o_tr = o_ar + (double)vr;
        o_tc = o_ac + (double)vc;
      } else {
        double x_r = (double)r_pen, x_c = (double)c;
        if (pe) {
          const float pb = pc * bc;
          const double m1 = d_g * (double)(br - pb);
          x_r = x_r + m1;
          const double m2 = d_g * (double)bc;
          x_c = x_c + m2;
        }
        o_tr = osa_affine_scan64(x_r, resets, pw_g, crtg_r[q], lane);
        o_tc = osa_affine_scan64(x_c, resets, pw_g, crtg_c[q], lane);
        crtg_r[q] = osa_readlane63_f64(o_tr);
        crtg_c[q] = osa_readlane63_f64(o_tc);
      }
      cret[q] = osa_readlane63_f64(ret);
      cnv_r[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vr), 63));
      cnv_c[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vc), 63));
      s_out[0][lane][e] = (float)o_ar;
      s_out[1][lane][e] = (float)o_ac;
      s_out[2][lane][e] = (float)o_tr;
      s_out[3][lane][e] = (float)o_tc;
      s_out[4][lane][e] = (float)ret;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + NTH * k, tt = i / NB, e = i % NB;
      const int t = t_hi - tt, n = n0 + e;
      if (tt < nvalid && n < N) {
        const long g = (long)t * N + n;
        adv_r[g] = s_out[0][tt][e];
        adv_c[g] = s_out[1][tt][e];
        tgt_r[g] = s_out[2][tt][e];
        tgt_c[g] = s_out[3][tt][e];
        disc_ret[g] = s_out[4][tt][e];
      }
    }
    if (j + 1 < ntiles) stage();
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// K5c  the dual GAE scan SPLIT OVER TIME across workgroups (round 3): "chained" lane-per-env scan.
//
// The lane-per-env kernel (K5) does the minimum arithmetic per transition but walks T sequentially per lane: it
// needs N >= ~0.5 M envs to keep the chip's memory system busy (4.8 TB/s at 16 x 1 M, 2.4 TB/s at 256 x 65 536,
// profiles/r2_gae_bandwidth.md).  The time-parallel kernel (K5b) fills lanes with time steps but pays a log-depth
// float64 DPP scan per tile (2.5 TB/s at 4096 x 4096).  Here the (T, N) buffer is cut into LEVELS of 128 steps:
// a workgroup = 8 waves x 64 envs, wave w owns 16 consecutive steps of the level and keeps its INPUTS in
// registers (112 loads in flight per lane: the whole chunk is requested before anything is consumed), so the chip
// runs (T / 16) x (N / 64) waves regardless of the shape.  Every recurrence y_t = x_t + d y_{t+1} is affine in
// its incoming carry (coefficient d^len, or 0 once a path ends inside the stretch), so
//   1. pass 1: each wave runs the sequential float64 recurrences over its 16 steps with ZERO carries -> its
//      aggregate (the carries it would hand on) and an "open" bit (no path end in the chunk);
//   2. the waves' aggregates meet in LDS; the level's aggregate is published (agent-scope 8-byte words, NaN
//      sentinel = not yet there; a level with a path end publishes its INCLUSIVE carry right away: it does not
//      depend on what comes in, which cuts the chain at every episode boundary);
//   3. decoupled look-back (lane-wise: every env has its own chain): walk to later levels until one has its
//      inclusive carry, then fold the aggregates back in level order, incl_j = agg_j + d^128 incl_{j-1} -- a pure
//      function of the aggregates, so the bits do not depend on timing; publish the own inclusive carry;
//   4. pass 2: each wave folds the level's incoming carry through the waves before it and re-runs the sequential
//      recurrences FROM REGISTERS with its true incoming carries, writing the outputs.
// One read of the inputs, one write of the outputs, + ~0.4 B per transition of carries.  Levels are mapped to
// block indices latest-first, so a workgroup only ever waits for blocks dispatched before it.  Arithmetic: the
// sequential kernel's, step for step, given the incoming carry; the carry itself is assembled by the affine
// identity instead of the chain (float64 re-association: outputs equal the bit-exact kernel's in all but
// ~1e-8 of the elements after rounding to float32; tests require rtol 1e-5 like K5b).  v-trace: K5 only.
// ------------------------------------------------------------------------------------------------
#ifndef OSA_GC_TC
#define OSA_GC_TC 16                       // steps per wave
#endif
#ifndef OSA_GC_NW
#define OSA_GC_NW 8                        // waves per workgroup
#endif
#define OSA_GC_LEV (OSA_GC_TC * OSA_GC_NW)  // steps per level

struct OsaGaeCarry {
  double v[5];  // a_r, a_c, ret, rtg_r, rtg_c
};

// "not there yet" = the all-ones pattern the workspace is pre-set to (a NaN no arithmetic produces: genuine NaNs
// of a diverged run have other payloads, count as values and propagate exactly as in the sequential kernel)
__device__ __forceinline__ bool osa_gc_ready(double x) { return __double_as_longlong(x) != -1ll; }
__device__ __forceinline__ void osa_gc_put(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double osa_gc_get(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(
      reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// bounded (NaN inputs / a preempted device must not hang it); giving up raises the STICKY word `tmo` -- the carry
// returned then is the sentinel itself, i.e. NaN outputs, and the host turns the word into an error at its next
// synchronisation (osa_gae_chained_timed_out) instead of training on them silently
__device__ __forceinline__ double osa_gc_wait(const double* p, unsigned int* tmo) {
  double v = osa_gc_get(p);
  for (int spins = 0; !osa_gc_ready(v); ++spins) {
    if (spins >= (1 << 22)) {
      __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
    v = osa_gc_get(p);
  }
  return v;
}

// the sequential kernel's step (K5, without v-trace) over one 16-step chunk held in registers
template <int EST, bool STORE>
__device__ __forceinline__ void osa_gc_chunk(const float (&r)[OSA_GC_TC], const float (&c)[OSA_GC_TC],
                                             const float (&vr)[OSA_GC_TC], const float (&vc)[OSA_GC_TC],
                                             const float (&br)[OSA_GC_TC], const float (&bc)[OSA_GC_TC],
                                             const uint8_t (&e)[OSA_GC_TC], int nsteps, float nv_r, float nv_c,
                                             OsaGaeCarry& cy, bool& open, float g32, double d_g, double d_r,
                                             double d_c, float pc, long i_hi, int N, float* __restrict__ adv_r,
                                             float* __restrict__ adv_c, float* __restrict__ tgt_r,
                                             float* __restrict__ tgt_c, float* __restrict__ disc_ret) {
#pragma clang fp contract(off)
  double a_r = cy.v[0], a_c = cy.v[1], ret = cy.v[2], rtg_r = cy.v[3], rtg_c = cy.v[4];
#pragma unroll
  for (int u = 0; u < OSA_GC_TC; ++u) {
    if (u >= nsteps) continue;  // wave-uniform (no `break`: the loop must unroll, the chunk lives in registers)
    if (e[u]) {  // a path ends after this step: re-seed every carry with the bootstrap
      nv_r = br[u];
      nv_c = bc[u];
      a_r = 0.0;
      a_c = 0.0;
      ret = (double)br[u];
      const float pb = pc * bc[u];
      rtg_r = (double)(br[u] - pb);
      rtg_c = (double)bc[u];
      open = false;
    }
    const float pcost = pc * c[u];
    const float r_pen = r[u] - pcost;
    const float gr = g32 * nv_r;
    const float gc = g32 * nv_c;
    const float sr = r_pen + gr;
    const float sc = c[u] + gc;
    const float delta_r = sr - vr[u];
    const float delta_c = sc - vc[u];
    const double m0 = d_g * ret;
    ret = (double)r[u] + m0;
    double out_ar, out_ac, out_tr, out_tc;
    if (EST != OSA_EST_GAE) {
      const double m1 = d_g * rtg_r;
      rtg_r = (double)r_pen + m1;
      const double m2 = d_g * rtg_c;
      rtg_c = (double)c[u] + m2;
    }
    if (EST == OSA_EST_PLAIN) {
      out_ar = (double)delta_r;
      out_ac = (double)delta_c;
    } else {
      const double m3 = d_r * a_r;
      a_r = (double)delta_r + m3;
      const double m4 = d_c * a_c;
      a_c = (double)delta_c + m4;
      out_ar = a_r;
      out_ac = a_c;
    }
    if (EST == OSA_EST_GAE) {
      out_tr = out_ar + (double)vr[u];
      out_tc = out_ac + (double)vc[u];
    } else {
      out_tr = rtg_r;
      out_tc = rtg_c;
    }
    if (STORE) {
      const long i = i_hi - (long)u * N;
      adv_r[i] = (float)out_ar;
      adv_c[i] = (float)out_ac;
      tgt_r[i] = (float)out_tr;
      tgt_c[i] = (float)out_tc;
      disc_ret[i] = (float)ret;
    }
    nv_r = vr[u];
    nv_c = vc[u];
  }
  cy.v[0] = a_r; cy.v[1] = a_c; cy.v[2] = ret; cy.v[3] = rtg_r; cy.v[4] = rtg_c;
}

#ifndef OSA_GC_MINB
#define OSA_GC_MINB 1  // (2 = two workgroups per compute unit: 159 v 165 us at 4096 x 4096 in one same-box A/B, nothing on other shapes: inside the box-to-box noise, not adopted)
#endif
template <int EST>
__global__ __launch_bounds__(64 * OSA_GC_NW, OSA_GC_MINB) void osa_gae_chain_scan_kernel(
    const float* __restrict__ reward, const float* __restrict__ cost,
    const float* __restrict__ value_r, const float* __restrict__ value_c,
    const uint8_t* __restrict__ path_end, const float* __restrict__ boot_r,
    const float* __restrict__ boot_c, int T, int N, float g32, double d_g, double d_r, double d_c,
    float pc, float* __restrict__ adv_r, float* __restrict__ adv_c, float* __restrict__ tgt_r,
    float* __restrict__ tgt_c, float* __restrict__ disc_ret, double* __restrict__ ws, int nenvb,
    unsigned int* __restrict__ ticket) {
#pragma clang fp contract(off)
  constexpr int TC = OSA_GC_TC, NW = OSA_GC_NW;
  // which carries exist: GAE a_r a_c ret | GAE-RTG all five | PLAIN ret rtg_r rtg_c; KMASK bit k = carry k is live
  constexpr int KMASK = EST == OSA_EST_GAE ? 0b00111 : (EST == OSA_EST_GAE_RTG ? 0b11111 : 0b11100);
  __shared__ double s_agg[NW][5][64];
  __shared__ double s_cin[5][64];
  __shared__ uint8_t s_open[NW][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // logical block index = order of ARRIVAL (a ticket), not blockIdx: a workgroup then only ever waits for
  // workgroups that are already running or done, whatever order the dispatcher picks (rocPRIM's look-back scan
  // does the same)
  __shared__ int s_bid;
  if (threadIdx.x == 0)
    s_bid = (int)__hip_atomic_fetch_add(reinterpret_cast<unsigned int*>(ticket), 1u, __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int bid = s_bid;
  const int lev = bid / nenvb, eb = bid - lev * nenvb;  // level 0 = the LAST 128 steps of the buffer
  const int n = eb * 64 + lane;
  const bool live = n < N;
  const int nc = live ? n : N - 1;  // dead lanes read a valid column and never publish or store
  const int t_hi = T - 1 - lev * OSA_GC_LEV - wave * TC;
  const int nsteps = t_hi < 0 ? 0 : (t_hi + 1 < TC ? t_hi + 1 : TC);  // wave-uniform
  const double dk[5] = {d_r, d_c, d_g, d_g, d_g};
  double p16[5], p128[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    p16[k] = osa_powi(dk[k], TC);
    p128[k] = osa_powi(p16[k], NW);  // d^(steps per level)
  }
  // ---- the chunk's inputs: everything requested before anything is consumed
  float r[TC], c[TC], vr[TC], vc[TC], br[TC], bc[TC];
  uint8_t e[TC];
  const long i_hi = (long)(t_hi < 0 ? 0 : t_hi) * N + nc;
#pragma unroll
  for (int u = 0; u < TC; ++u) e[u] = path_end[(u < nsteps) ? i_hi - (long)u * N : i_hi];  // (first: see below)
#pragma unroll
  for (int u = 0; u < TC; ++u) {
    const long i = (u < nsteps) ? i_hi - (long)u * N : i_hi;
    r[u] = reward[i];
    c[u] = cost[i];
    vr[u] = value_r[i];
    vc[u] = value_c[i];
  }
  // the bootstraps only count where a path ends: cache lines without any path end (most of them) are never
  // read -- up to 8 of the 45 bytes a transition would otherwise move.  The flags were requested first, so this
  // wait leaves the 64 loads above in flight.
#pragma unroll
  for (int u = 0; u < TC; ++u) {  // (lane-predicated loads, as K5: lanes without a path end generate no traffic)
    const long i = (u < nsteps) ? i_hi - (long)u * N : i_hi;
    br[u] = e[u] ? boot_r[i] : 0.f;
    bc[u] = e[u] ? boot_c[i] : 0.f;
  }
  // value of the step after the chunk (the sequential kernel's running nv; 0 past the end of the buffer)
  float nv_r = 0.f, nv_c = 0.f;
  if (nsteps > 0 && t_hi + 1 < T) {
    nv_r = value_r[i_hi + N];
    nv_c = value_c[i_hi + N];
  }
  // ---- pass 1: aggregate of the chunk (zero incoming carries)
  OsaGaeCarry agg = {{0.0, 0.0, 0.0, 0.0, 0.0}};
  bool open = true;
  osa_gc_chunk<EST, false>(r, c, vr, vc, br, bc, e, nsteps, nv_r, nv_c, agg, open, g32, d_g, d_r, d_c, pc, i_hi, N,
                           adv_r, adv_c, tgt_r, tgt_c, disc_ret);
#pragma unroll
  for (int k = 0; k < 5; ++k)
    if ((KMASK >> k) & 1) s_agg[wave][k][lane] = agg.v[k];
  s_open[wave][lane] = open ? 1 : 0;
  __syncthreads();
  // ---- the level's aggregate, look-back, the level's incoming carry (wave 0; one chain per lane = env)
  const int nlevslots = 10;  // per level: [agg 0..4 | incl 0..4][N] doubles
  if (wave == 0) {
    double lagg[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    bool lopen = true;
#pragma unroll
    for (int w = 0; w < NW; ++w) {  // wave 0 holds the latest steps: carries flow from wave 0 to wave NW - 1
      const bool ow = s_open[w][lane] != 0;
#pragma unroll
      for (int k = 0; k < 5; ++k)
        if ((KMASK >> k) & 1) {
          // (a select, not a product with 0: a NaN behind a path end must stop there, as in the sequential kernel)
          const double m = p16[k] * lagg[k];
          const double av = s_agg[w][k][lane];
          lagg[k] = ow ? av + m : av;
        }
      lopen = lopen && ow;
    }
    double cin[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    double* mine = ws + ((long)lev * nlevslots) * N + nc;
    if (lev == 0 || !lopen) {  // the inclusive carry of this level does not depend on what comes in: publish now
      if (live) {
#pragma unroll
        for (int k = 0; k < 5; ++k)
          if ((KMASK >> k) & 1) osa_gc_put(mine + (long)(5 + k) * N, lagg[k]);
      }
    } else if (live) {
#pragma unroll
      for (int k = 0; k < 5; ++k)
        if ((KMASK >> k) & 1) osa_gc_put(mine + (long)k * N, lagg[k]);
    }
    if (lev > 0 && live) {
      constexpr int K0 = (KMASK & 1) ? 0 : 2;  // a carry every estimator has (polled first)
      // walk to later levels until one has published its inclusive carry (a level with a path end always has)
      int j = lev - 1;
      unsigned int* tmo = ticket + 2;  // sticky time-out word: the double behind the ticket's
      for (int spins = 0;; ++spins) {  // (bounded: never hang the device; the waits below are too)
        if (spins >= (1 << 22)) {
          __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        const double* lv = ws + ((long)j * nlevslots) * N + nc;
        double vi = osa_gc_get(lv + (long)(5 + K0) * N);
        if (osa_gc_ready(vi)) break;
        const double va = osa_gc_get(lv + (long)K0 * N);
        if (osa_gc_ready(va) && j > 0) { --j; continue; }  // open level, aggregate there: look further (level 0 always publishes incl)
        __builtin_amdgcn_s_sleep(1);
      }
      {  // inclusive carry of level j, then fold the aggregates of levels j + 1 .. lev - 1 back in, in level order
        const double* lv = ws + ((long)j * nlevslots) * N + nc;
#pragma unroll
        for (int k = 0; k < 5; ++k)
          if ((KMASK >> k) & 1) cin[k] = osa_gc_wait(lv + (long)(5 + k) * N, tmo);
      }
      for (int q = j + 1; q < lev; ++q) {
        const double* lv = ws + ((long)q * nlevslots) * N + nc;
#pragma unroll
        for (int k = 0; k < 5; ++k)
          if ((KMASK >> k) & 1) {
            const double m = p128[k] * cin[k];
            cin[k] = osa_gc_wait(lv + (long)k * N, tmo) + m;
          }
      }
      if (lopen) {  // this level's inclusive carry, by the same formula anybody else would use for it
#pragma unroll
        for (int k = 0; k < 5; ++k)
          if ((KMASK >> k) & 1) {
            const double m = p128[k] * cin[k];
            osa_gc_put(mine + (long)(5 + k) * N, lagg[k] + m);
          }
      }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) s_cin[k][lane] = cin[k];
  }
  __syncthreads();
  // ---- pass 2: this wave's incoming carries (the level's, folded through the waves before it), final outputs
  OsaGaeCarry cy;
#pragma unroll
  for (int k = 0; k < 5; ++k) cy.v[k] = s_cin[k][lane];
  for (int w = 0; w < wave; ++w) {
    const bool ow = s_open[w][lane] != 0;
#pragma unroll
    for (int k = 0; k < 5; ++k)
      if ((KMASK >> k) & 1) {
        const double m = p16[k] * cy.v[k];
        const double av = s_agg[w][k][lane];
        cy.v[k] = ow ? av + m : av;
      }
  }
  if (!live || nsteps == 0) return;
  bool dummy = true;
  osa_gc_chunk<EST, true>(r, c, vr, vc, br, bc, e, nsteps, nv_r, nv_c, cy, dummy, g32, d_g, d_r, d_c, pc, i_hi, N,
                          adv_r, adv_c, tgt_r, tgt_c, disc_ret);
}

// ------------------------------------------------------------------------------------------------
// K6  statistics: deterministic two-stage float64 reductions
// ------------------------------------------------------------------------------------------------
#define OSA_RED_BLOCKS 512
#define OSA_RED_THREADS 256

__global__ __launch_bounds__(OSA_RED_THREADS) void osa_sum2_partial_kernel(
    const float* __restrict__ a, const float* __restrict__ b, long M, double* __restrict__ ws) {
  __shared__ double red[17];
  double sa = 0.0, sb = 0.0;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += stride) {
    sa += (double)a[i];
    sb += (double)b[i];
  }
  sa = osa_block_sum<OSA_RED_THREADS>(sa, red);
  sb = osa_block_sum<OSA_RED_THREADS>(sb, red);
  if (threadIdx.x == 0) {
    ws[2 * blockIdx.x] = sa;
    ws[2 * blockIdx.x + 1] = sb;
  }
}

__global__ __launch_bounds__(OSA_RED_THREADS) void osa_sum2_final_kernel(
    const double* __restrict__ ws, int nblk, long M, double* __restrict__ stats) {
  __shared__ double red[17];
  double sa = 0.0, sb = 0.0;
  for (int i = threadIdx.x; i < nblk; i += OSA_RED_THREADS) {
    sa += ws[2 * i];
    sb += ws[2 * i + 1];
  }
  sa = osa_block_sum<OSA_RED_THREADS>(sa, red);
  sb = osa_block_sum<OSA_RED_THREADS>(sb, red);
  if (threadIdx.x == 0) {
    stats[0] = sa;
    stats[1] = sb;
    stats[2] = (double)M;
  }
}

__global__ __launch_bounds__(OSA_RED_THREADS) void osa_sumsq_partial_kernel(
    const float* __restrict__ a, long M, const double* __restrict__ stats,
    double* __restrict__ ws) {
#pragma clang fp contract(off)
  __shared__ double red[17];
  // float32 mean exactly as the reference forms it: float32 sum / n (distributed.py:384)
  const float mean = (float)stats[0] / (float)stats[2];
  double s = 0.0;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += stride) {
    const float d = a[i] - mean;
    const float q = d * d;
    s += (double)q;
  }
  s = osa_block_sum<OSA_RED_THREADS>(s, red);
  if (threadIdx.x == 0) ws[blockIdx.x] = s;
}

__global__ __launch_bounds__(OSA_RED_THREADS) void osa_sumsq_final_kernel(
    const double* __restrict__ ws, int nblk, double* __restrict__ stats) {
  __shared__ double red[17];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += OSA_RED_THREADS) s += ws[i];
  s = osa_block_sum<OSA_RED_THREADS>(s, red);
  if (threadIdx.x == 0) {
    stats[3] = s;
    stats[4] = (double)((float)stats[0] / (float)stats[2]);
    stats[5] = (double)((float)stats[1] / (float)stats[2]);
  }
}

// ------------------------------------------------------------------------------------------------
// K6  get(): (T, N) -> (N*T) transposition through a 64x64 LDS tile, standardisation fused
// ------------------------------------------------------------------------------------------------
struct OsaGetScalars {
  const float* src[6];
  float* dst[6];
  int op[6];  // 0 copy, 1 (x - mean_r)/(std_r + 1e-8), 2 x - mean_c
};

__global__ __launch_bounds__(256) void osa_get_scalars_kernel(OsaGetScalars p, int T, int N,
                                                              double* __restrict__ stats) {
#pragma clang fp contract(off)
  __shared__ float tile[64][65];
  const int n0 = blockIdx.x * 64, t0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  const float mean_r = (float)stats[4], mean_c = (float)stats[5];
  const float std_r = sqrtf((float)stats[3] / (float)stats[2]);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) stats[6] = (double)std_r;
  const float denom = std_r + 1e-8f;
  // Round 6: the six arrays' elements of this thread are requested TOGETHER (clamped addresses, up to 96 loads in
  // flight), then transposed array by array: one memory round trip instead of six behind one another -- at the
  // headline shape (16 x 4096) the launch is nothing but those trips (22 us)
  float pre[6][16];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const bool have = p.src[k] != nullptr && p.dst[k] != nullptr;
    const float* __restrict__ src = have ? p.src[k] : reinterpret_cast<const float*>(stats);  // (any readable address)
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int t = t0 + ty + 4 * u, n = n0 + tx;
      const bool ok = have && t < T && n < N;
      pre[k][u] = src[ok ? (long)t * N + n : 0];
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float* __restrict__ src = p.src[k];
    float* __restrict__ dst = p.dst[k];
    if (src == nullptr || dst == nullptr) continue;
    const int op = p.op[k];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int tt = ty + 4 * u;
      const int t = t0 + tt, n = n0 + tx;
      if (t < T && n < N) {
        float v = pre[k][u];
        if (op == 1) v = (v - mean_r) / denom;
        else if (op == 2) v = v - mean_c;
        tile[tt][tx] = v;
      }
    }
    __syncthreads();
    for (int nn = ty; nn < 64; nn += 4) {
      const int n = n0 + nn, t = t0 + tx;
      if (t < T && n < N) dst[(long)n * T + t] = tile[tx][nn];
    }
  }
}

__global__ __launch_bounds__(256) void osa_get_rows_kernel(const float* __restrict__ src, int ld_src,
                                                           float* __restrict__ dst, int ld_dst,
                                                           int T, int N, int dim) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)T * N * dim;
  if (gid >= total) return;
  const long row = gid / dim;  // env-major sample index i = n*T + t
  const int d = (int)(gid - row * dim);
  const int n = (int)(row / T), t = (int)(row - (long)n * T);
  dst[row * ld_dst + d] = src[((long)t * N + n) * ld_src + d];
}

// The same re-ordering with 16-byte accesses and 32-bit index arithmetic (rows 16-byte aligned, dim % 4 == 0,
// fewer than 2^31 vectors): one thread moves one float4; consecutive threads walk the DESTINATION (contiguous
// stores; the loads are row segments N * ld_src apart).  The scalar kernel above spends two 64-bit divisions per
// 4 bytes and reached 2.3 TB/s at 16 M rows of 60 floats.
__global__ __launch_bounds__(256) void osa_get_rows_vec4_kernel(const float* __restrict__ src, int ld_src,
                                                                float* __restrict__ dst, int ld_dst,
                                                                unsigned T, unsigned N, unsigned dim4,
                                                                unsigned total4) {
  const unsigned gid = blockIdx.x * 256u + threadIdx.x;
  if (gid >= total4) return;
  const unsigned row = gid / dim4, q = gid - row * dim4;  // env-major sample index i = n*T + t
  const unsigned n = row / T, t = row - n * T;
  const float4 v = *reinterpret_cast<const float4*>(src + ((size_t)t * N + n) * ld_src + 4 * q);
  *reinterpret_cast<float4*>(dst + (size_t)row * ld_dst + 4 * q) = v;
}

static void osa_launch_get_rows(const float* src, int ld_src, float* dst, int ld_dst, int T, int N, int dim,
                                hipStream_t st) {
  const long total = (long)T * N * dim;
  const bool vec = dim % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && total / 4 < 2147483647L &&
                   (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
  if (vec) {
    const unsigned total4 = (unsigned)(total / 4);
    hipLaunchKernelGGL(osa_get_rows_vec4_kernel, dim3((total4 + 255) / 256), dim3(256), 0, st, src, ld_src, dst,
                       ld_dst, (unsigned)T, (unsigned)N, (unsigned)(dim / 4), total4);
  } else {
    hipLaunchKernelGGL(osa_get_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, ld_src,
                       dst, ld_dst, T, N, dim);
  }
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* osa_strerror(int code) {
  switch (code) {
    case OSA_OK: return "ok";
    case OSA_EINVAL: return "invalid argument";
    case OSA_EHIP: return "HIP runtime error";
    case OSA_EUNSUPPORTED: return "not implemented in libomnisafe_amd";
    default: return "unknown error";
  }
}
int osa_version(void) { return 1; }
const char* osa_build_arch(void) { return "gfx950"; }

int osa_buffer_store_step(int t, int N, int obs_dim, int act_dim, const float* obs, int ld_obs_src,
                          const float* act, int ld_act_src, const float* reward, const float* cost,
                          const float* value_r, const float* value_c, const float* logp,
                          float* buf_obs, int ld_obs_buf, float* buf_act, int ld_act_buf,
                          float* buf_reward, float* buf_cost, float* buf_value_r, float* buf_value_c,
                          float* buf_logp, void* stream) {
  OSA_REQUIRE(t >= 0 && N > 0 && obs_dim > 0 && act_dim > 0);
  OSA_REQUIRE(obs && act && reward && cost && value_r && value_c && logp);
  OSA_REQUIRE(buf_obs && buf_act && buf_reward && buf_cost && buf_value_r && buf_value_c && buf_logp);
  OSA_REQUIRE(ld_obs_src >= obs_dim && ld_obs_buf >= obs_dim && ld_act_src >= act_dim &&
              ld_act_buf >= act_dim);
  const long row = (long)t * N;
  const long work = (long)N * (obs_dim > act_dim ? obs_dim : act_dim);
  const int blocks = (int)((work + 255) / 256);
  hipLaunchKernelGGL(osa_store_step_kernel, dim3(blocks), dim3(256), 0, osa_stream(stream), N,
                     obs_dim, act_dim, obs, ld_obs_src, act, ld_act_src, reward, cost, value_r,
                     value_c, logp, buf_obs + row * ld_obs_buf, ld_obs_buf,
                     buf_act + row * ld_act_buf, ld_act_buf, buf_reward + row, buf_cost + row,
                     buf_value_r + row, buf_value_c + row, buf_logp + row);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_gae_scan(const float* reward, const float* cost, const float* value_r, const float* value_c,
                 const uint8_t* path_end, const float* boot_r, const float* boot_c, int T, int N,
                 double gamma, double lam, double lam_c, float penalty_coef, int estimator,
                 float* adv_r, float* adv_c, float* target_value_r, float* target_value_c,
                 float* discounted_ret, void* stream) {
  OSA_REQUIRE(T > 0 && N > 0);
  OSA_REQUIRE(reward && cost && value_r && value_c && path_end && boot_r && boot_c);
  OSA_REQUIRE(adv_r && adv_c && target_value_r && target_value_c && discounted_ret);
  if (estimator < OSA_EST_GAE || estimator > OSA_EST_VTRACE) return OSA_EUNSUPPORTED;
  const float g32 = (float)gamma;  // gamma * float32 tensor: the scalar is rounded to float32
  const double d_g = gamma, d_r = gamma * lam, d_c = gamma * lam_c;  // python-float products
  const int threads = N >= 256 ? 256 : 64;
  const int blocks = (N + threads - 1) / threads;
  constexpr int U = 8;
#define OSA_GAE_LAUNCH(E)                                                                         \
  hipLaunchKernelGGL((osa_gae_scan_kernel<E, U>), dim3(blocks), dim3(threads), 0,                 \
                     osa_stream(stream), reward, cost, value_r, value_c, path_end, boot_r, boot_c, \
                     T, N, g32, d_g, d_r, d_c, penalty_coef, adv_r, adv_c, target_value_r,        \
                     target_value_c, discounted_ret)
  if (estimator == OSA_EST_GAE) OSA_GAE_LAUNCH(OSA_EST_GAE);
  else if (estimator == OSA_EST_GAE_RTG) OSA_GAE_LAUNCH(OSA_EST_GAE_RTG);
  else if (estimator == OSA_EST_PLAIN) OSA_GAE_LAUNCH(OSA_EST_PLAIN);
  else OSA_GAE_LAUNCH(OSA_EST_VTRACE);
#undef OSA_GAE_LAUNCH
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_gae_scan_tiled(const float* reward, const float* cost, const float* value_r, const float* value_c,
                       const uint8_t* path_end, const float* boot_r, const float* boot_c, int T, int N,
                       double gamma, double lam, double lam_c, float penalty_coef, int estimator,
                       float* adv_r, float* adv_c, float* target_value_r, float* target_value_c,
                       float* discounted_ret, void* stream) {
  OSA_REQUIRE(T > 0 && N > 0);
  OSA_REQUIRE(reward && cost && value_r && value_c && path_end && boot_r && boot_c);
  OSA_REQUIRE(adv_r && adv_c && target_value_r && target_value_c && discounted_ret);
  if (estimator < OSA_EST_GAE || estimator > OSA_EST_PLAIN) return OSA_EUNSUPPORTED;  // v-trace: float32 chain
  const float g32 = (float)gamma;
  const double d_g = gamma, d_r = gamma * lam, d_c = gamma * lam_c;
  constexpr int NB = 16;
  const int blocks = (N + NB - 1) / NB;
  // 16 waves per workgroup (one env per wave, four waves per SIMD) while the grid is small enough to leave the
  // compute units under-occupied; OSA_GAE_TILE_WAVES = 4 / 16 forces one (A/B switch of tools/gae_bandwidth.py)
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  bool wide_wg = blocks <= 2 * cus;
  if (const char* e = getenv("OSA_GAE_TILE_WAVES")) wide_wg = e[0] == '1';
#define OSA_GAE_TILE_LAUNCH(E)                                                                            \
  do {                                                                                                    \
    if (wide_wg)                                                                                          \
      hipLaunchKernelGGL((osa_gae_tile_scan_kernel<E, NB, 16>), dim3(blocks), dim3(1024), 0,              \
                         osa_stream(stream), reward, cost, value_r, value_c, path_end, boot_r, boot_c, T, \
                         N, g32, d_g, d_r, d_c, penalty_coef, adv_r, adv_c, target_value_r,               \
                         target_value_c, discounted_ret);                                                 \
    else                                                                                                  \
      hipLaunchKernelGGL((osa_gae_tile_scan_kernel<E, NB, 4>), dim3(blocks), dim3(256), 0,                \
                         osa_stream(stream), reward, cost, value_r, value_c, path_end, boot_r, boot_c, T, \
                         N, g32, d_g, d_r, d_c, penalty_coef, adv_r, adv_c, target_value_r,               \
                         target_value_c, discounted_ret);                                                 \
  } while (0)
  if (estimator == OSA_EST_GAE) OSA_GAE_TILE_LAUNCH(OSA_EST_GAE);
  else if (estimator == OSA_EST_GAE_RTG) OSA_GAE_TILE_LAUNCH(OSA_EST_GAE_RTG);
  else OSA_GAE_TILE_LAUNCH(OSA_EST_PLAIN);
#undef OSA_GAE_TILE_LAUNCH
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

size_t osa_gae_chained_ws_doubles(int T, int N) {
  if (T < 1 || N < 1) return 0;
  // carries + the arrival ticket + the sticky time-out word (the LAST double: zeroed by the caller once, never by a scan)
  return (size_t)((T + OSA_GC_LEV - 1) / OSA_GC_LEV) * 10 * (size_t)N + 2;
}

int osa_gae_chained_timed_out(const double* ws, int T, int N, int* out) {
  OSA_REQUIRE(ws && out && T > 0 && N > 0);
  const size_t nws = osa_gae_chained_ws_doubles(T, N);
  return hipMemcpy(out, ws + (nws - 1), sizeof(int), hipMemcpyDeviceToHost) == hipSuccess ? OSA_OK : OSA_EHIP;
}

int osa_gae_scan_chained(const float* reward, const float* cost, const float* value_r, const float* value_c,
                         const uint8_t* path_end, const float* boot_r, const float* boot_c, int T, int N,
                         double gamma, double lam, double lam_c, float penalty_coef, int estimator,
                         float* adv_r, float* adv_c, float* target_value_r, float* target_value_c,
                         float* discounted_ret, double* ws, void* stream) {
  OSA_REQUIRE(T > 0 && N > 0 && ws);
  OSA_REQUIRE(reward && cost && value_r && value_c && path_end && boot_r && boot_c);
  OSA_REQUIRE(adv_r && adv_c && target_value_r && target_value_c && discounted_ret);
  if (estimator < OSA_EST_GAE || estimator > OSA_EST_PLAIN) return OSA_EUNSUPPORTED;  // v-trace: float32 chain
  const float g32 = (float)gamma;
  const double d_g = gamma, d_r = gamma * lam, d_c = gamma * lam_c;
  const int nlev = (T + OSA_GC_LEV - 1) / OSA_GC_LEV, nenvb = (N + 63) / 64;
  if ((long)nlev * nenvb > 2147483647L) return OSA_EUNSUPPORTED;
  hipStream_t st = osa_stream(stream);
  // every carry slot starts as the "not yet there" NaN sentinel
  const size_t nws = osa_gae_chained_ws_doubles(T, N);
  if (hipMemsetAsync(ws, 0xFF, (nws - 2) * sizeof(double), st) != hipSuccess) return OSA_EHIP;
  unsigned int* ticket = reinterpret_cast<unsigned int*>(ws + (nws - 2));  // (ticket[2] = the sticky time-out word)
  if (hipMemsetAsync(ticket, 0, sizeof(double), st) != hipSuccess) return OSA_EHIP;
#define OSA_GAE_CHAIN_LAUNCH(E)                                                                             \
  hipLaunchKernelGGL((osa_gae_chain_scan_kernel<E>), dim3(nlev * nenvb), dim3(64 * OSA_GC_NW), 0, st, reward, \
                     cost, value_r, value_c, path_end, boot_r, boot_c, T, N, g32, d_g, d_r, d_c, penalty_coef, \
                     adv_r, adv_c, target_value_r, target_value_c, discounted_ret, ws, nenvb, ticket)
  if (estimator == OSA_EST_GAE) OSA_GAE_CHAIN_LAUNCH(OSA_EST_GAE);
  else if (estimator == OSA_EST_GAE_RTG) OSA_GAE_CHAIN_LAUNCH(OSA_EST_GAE_RTG);
  else OSA_GAE_CHAIN_LAUNCH(OSA_EST_PLAIN);
#undef OSA_GAE_CHAIN_LAUNCH
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

size_t osa_reduce_ws_bytes(void) { return (size_t)OSA_RED_BLOCKS * 2 * sizeof(double); }

static int osa_red_blocks(long M) {
  long b = (M + (long)OSA_RED_THREADS * 4 - 1) / ((long)OSA_RED_THREADS * 4);
  if (b < 1) b = 1;
  if (b > OSA_RED_BLOCKS) b = OSA_RED_BLOCKS;
  return (int)b;
}

int osa_adv_stats_phase1(const float* adv_r, const float* adv_c, long M, double* ws, double* stats,
                         void* stream) {
  OSA_REQUIRE(adv_r && adv_c && ws && stats && M > 0);
  const int nblk = osa_red_blocks(M);
  hipLaunchKernelGGL(osa_sum2_partial_kernel, dim3(nblk), dim3(OSA_RED_THREADS), 0,
                     osa_stream(stream), adv_r, adv_c, M, ws);
  hipLaunchKernelGGL(osa_sum2_final_kernel, dim3(1), dim3(OSA_RED_THREADS), 0, osa_stream(stream),
                     ws, nblk, M, stats);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_adv_stats_phase2(const float* adv_r, long M, double* ws, double* stats, void* stream) {
  OSA_REQUIRE(adv_r && ws && stats && M > 0);
  const int nblk = osa_red_blocks(M);
  hipLaunchKernelGGL(osa_sumsq_partial_kernel, dim3(nblk), dim3(OSA_RED_THREADS), 0,
                     osa_stream(stream), adv_r, M, stats, ws);
  hipLaunchKernelGGL(osa_sumsq_final_kernel, dim3(1), dim3(OSA_RED_THREADS), 0, osa_stream(stream),
                     ws, nblk, stats);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

int osa_buffer_get(int T, int N, int obs_dim, int act_dim, const float* obs, int ld_obs,
                   const float* act, int ld_act, const float* logp, const float* target_value_r,
                   const float* target_value_c, const float* adv_r, const float* adv_c,
                   const float* discounted_ret, double* stats, int standardize_r, int standardize_c,
                   float* out_obs, int ld_out_obs, float* out_act, int ld_out_act, float* out_logp,
                   float* out_target_value_r, float* out_target_value_c, float* out_adv_r,
                   float* out_adv_c, float* out_discounted_ret, void* stream) {
  OSA_REQUIRE(T > 0 && N > 0 && stats);
  OsaGetScalars p;
  const float* srcs[6] = {logp, target_value_r, target_value_c, adv_r, adv_c, discounted_ret};
  float* dsts[6] = {out_logp, out_target_value_r, out_target_value_c, out_adv_r, out_adv_c,
                    out_discounted_ret};
  const int ops[6] = {0, 0, 0, standardize_r ? 1 : 0, standardize_c ? 2 : 0, 0};
  for (int k = 0; k < 6; ++k) {
    p.src[k] = srcs[k];
    p.dst[k] = dsts[k];
    p.op[k] = ops[k];
  }
  hipLaunchKernelGGL(osa_get_scalars_kernel, dim3((N + 63) / 64, (T + 63) / 64), dim3(256), 0,
                     osa_stream(stream), p, T, N, stats);
  if (obs && out_obs) {
    OSA_REQUIRE(obs_dim > 0 && ld_obs >= obs_dim && ld_out_obs >= obs_dim);
    osa_launch_get_rows(obs, ld_obs, out_obs, ld_out_obs, T, N, obs_dim, osa_stream(stream));
  }
  if (act && out_act) {
    OSA_REQUIRE(act_dim > 0 && ld_act >= act_dim && ld_out_act >= act_dim);
    osa_launch_get_rows(act, ld_act, out_act, ld_out_act, T, N, act_dim, osa_stream(stream));
  }
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

}  // extern "C"
