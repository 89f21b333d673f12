// Persistent PPO-Lag update pass: ONE launch performs a whole pass of PolicyGradient._update's inner
// loop (policy_gradient.py:366-382) -- `nmb` dependent minibatch optimiser steps -- for all three
// networks (blockIdx.x = network; one 256-thread workgroup = 4 waves = 64 samples each).
//
// Why: with the reference's batch_size = 64 the update is a chain of 40 960 dependent optimiser steps
// per epoch.  A launch per step spends its time on global-memory latency (weights, Adam state and
// gradients make several L2 round trips per step: measured 57 us/step, of which MFMA work is ~5 us).
// Here the network lives in the CU for the whole pass:
//   * parameters: LDS-resident master copy (padded rows, conflict-free MFMA fragment reads),
//     loaded once, written back once;
//   * Adam moments m, v: REGISTERS (each lane owns the ~37 parameters its weight-gradient MFMA tiles
//     produce), loaded once, written back once;
//   * gradients never leave registers: MFMA accumulator -> (+2*coef*w) -> block-reduced norm -> clip
//     -> Adam -> LDS master;
//   * the next minibatch's rows (gathered by the permutation) are prefetched into the registers of the
//     current one as soon as those are consumed.
// Arithmetic follows osa_mb_grad_kernel + osa_finalize_net (same fragment algebra; the 1-2-output layers
// run on the VALU here: summation-order differences of ~1e-7); tests compare both against the reference's
// golden vectors.
//
// What bounds the step on gfx950 (measured, profiles/HISTORY.md §6): float32 MFMAs and VALU instructions do
// not overlap (in-wave or across waves), so the step costs MFMA cycles + VALU cycles + waits; the kernel is
// register-bound (256 VGPRs + ~200 AGPRs), and every runtime branch between phases is a scheduling barrier.
// Modes of the same kernel: plain pass (grid 3), data-parallel gradient step (grid 3 x world, writes
// slabs), cooperative data-parallel pass (COOP: 3 x world resident workgroups exchanging gradients),
// partial gradients of one large minibatch (MULTI, grid 3 x nblk).
#pragma once
#include "mlp_device.h"

#ifndef OSA_PART_MARK  // (phase clocks of the balanced partial-gradient kernel: part_grad_kernel.hip, -DOSA_PART_CLOCKS)
#define OSA_PART_MARK(k) do { } while (0)
#endif

#ifndef PSLD
#define PSLD 68  // leading dimension (floats) of [feature][sample] tiles and of W2/W3 rows
#endif
#define PSPAD (PSLD - 64)
#define PNSTAT 16

struct OsaPassHp {
  float clip, entropy_coef, critic_norm_coef, max_grad_norm;
  float lr_actor, lr_critic, beta1, beta2, adam_eps;
  int use_critic_norm, use_max_grad_norm, use_cost;
};

struct OsaPassArgs {
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
  long M;            // rows in the pass
  int B;             // minibatch size; > 64 is processed as ceil(B/64) chunks; last minibatch may be smaller
  int nmb;           // minibatches in this launch
  const float* lagrange;
  OsaPassHp hp;
  int loss_kind;
  int nets_mask;
  float* stats;  // [nmb][PNSTAT]
  // data-parallel gradient mode (osa_ppo_dp_step): grid (3, dp_world); workgroup (net, rk) processes
  // minibatch mb0 of virtual rank rk (rows rk*M .. rk*M+M-1 of the all-gathered arrays, permutation
  // perm[rk*M ..]) and writes its locally clipped gradient to dp_slabs instead of applying Adam.
  float* dp_slabs;  // [3][dp_world][P + PNSTAT] or nullptr
  int dp_world;
  int mb0;          // first minibatch index processed by this launch
  // cooperative data-parallel pass (osa_ppo_dp_pass): grid (3, dp_world), ALL workgroups persistent over
  // the nmb steps.  Workgroup (net, rk) computes rank rk's locally clipped gradient, publishes it in
  // dp_slabs (exchange layout, double-buffered by step parity), waits on dp_sync[net] for its dp_world
  // peers, then sums the dp_world gradients in rank order and applies Adam to ITS OWN copy of the
  // network: all peers perform identical arithmetic, so their copies stay bit-identical and nothing
  // but gradients ever crosses between compute units.
  int* dp_sync;     // [4]: arrival counters of the three networks + sticky time-out flag; or nullptr
  // partial-gradient mode of the large-batch step (osa_pass_partial_grad): grid (3, part_stride); workgroup
  // (net, b) accumulates the 64-row chunks b, b + part_stride, ... of ONE minibatch in its registers and
  // writes the raw sum (no L2 term, no clip) to slab b of dp_slabs; slab reduce + clip/Adam follow.
  int part_stride;  // 0 = off
  // BALANCED partial-gradient mode (osa_ppo_part_kernel, round 4; part_stride = -1): a 1-D grid of G workgroups, the
  // 64-row chunk-tasks of ALL networks of the launch laid end to end (network by network in mask order, chunk by
  // chunk) and cut into G contiguous ranges of part_tpw tasks: workgroup w owns tasks [w part_tpw, (w + 1) part_tpw).
  // A range that crosses a network boundary is processed in two segments (weights reloaded, one slab per segment).
  // 16 384 rows x 3 networks = 768 tasks on 256 compute units: 3 tasks each, where the strided mode above ran
  // 64 workgroups x 4 chunks per network on 192 units (44.7 -> 26 us per launch, profiles/HISTORY.md §7.4).  Slab index of a
  // segment = w - (first workgroup of that network); dp_world = slab stride per network.
  int part_tpw;
  // extended actor surrogates (EXT instantiations; osa_surrogate_ext of the public header): per-sample
  // KL(pi_theta || pi_old) term, FOCOPS trust mask, P3O exact penalty
  const float* old_mean;     // [rows][ld_old_mean]
  int ld_old_mean;
  const float* old_log_std;  // [act_dim]
  float ext_kl_coef, ext_mask_eta, ext_ratio_scale, ext_cost_kappa, ext_cost_excess;
  // 1: dp_slabs lives in device memory allocated uncached (osa_dp_exchange_alloc): every access is served by
  // the device-coherent level, so the hand-off needs no L2 write-back / invalidate (~2.5k cycles per step)
  int dp_uncached;
  // 1: the `world` workgroups of a network are blocks net + 8 rk of a 1-D grid, i.e. they run on ONE XCC
  // (workgroup b of a grid lands on XCC b mod 8) and dp_slabs is ordinary memory served by that XCC's L2
  int dp_local;
  // cooperative CHUNK mode (osa_ppo_chunked_pass): the dp_world workgroups of a network are the 64-row chunks
  // of ONE minibatch of B <= 64 dp_world rows (same data, same permutation): every workgroup computes the raw
  // gradient of its chunk (scaled by 1 / rows of the whole minibatch), the sum over the chunks is clipped by ITS
  // norm and every workgroup applies the same Adam step -- the arithmetic of a single-process minibatch of B
  // rows, not the clip-then-average of the data-parallel mode
  int dp_chunk;
  // chunk mode UNDER data parallelism (osa_ppo_dp_chunked_pass): dp_ranks > 1 virtual ranks, each a group of
  // dp_world / dp_ranks chunk workgroups (peer p = rank * chunks + chunk) working on ITS rows (rank r: rows
  // r M .. of the all-gathered arrays, permutation perm[r M ..]).  Two hand-offs per step: (1) inside the rank group
  // as in chunk mode -- sum of the chunks, ITS norm, the rank's clip factor; (2) across ranks -- every chunk
  // workgroup publishes its share of the tiles of the rank's sum, all dp_world peers arrive, everybody adds the
  // dp_ranks rank sums x clip factor in rank order, / dp_ranks (clip-then-average), same Adam step everywhere.
  // dp_sync then is int[64]: [net] stage-2 arrivals, [3] sticky flag, [4..7] placement, [8 + 16 net + rank] stage 1.
  int dp_ranks;
  // plain pass (grid 3): 1 = the three networks' workgroups are blocks 0, 8, 16 of a 17-block grid, i.e. (block b
  // runs on XCC b mod 8) they share ONE XCC and its L2: the rows all three gather (observations: 240 of the 268
  // bytes of a sample) are then fetched from HBM once instead of three times
  int one_xcc;
  long long* dbg;  // optional [3][16] accumulated phase cycles (s_memtime), or nullptr
  // ONE-SHOT PEER EXCHANGE (osa_ppo_p2p_pass; the P2P instantiations, round 6): real data parallelism without a
  // collective on the step path.  This process is rank p2p_rank of dp_world; it runs the single-GPU form of the pass
  // (3 workgroups, its OWN rows) and after the local clip every network's workgroup WRITES its gradient slab into the
  // exchange buffer of every rank (p2p_peer[q]: rank q's buffer as mapped into this process -- hipIpcOpenMemHandle;
  // [p2p_rank] is the rank's own allocation), releases at system scope, stores the step's sequence number into its
  // arrival word in every buffer, waits until all dp_world words of its OWN buffer carry the number, and sums the
  // slabs of its own buffer in rank order: identical arithmetic on every rank => bit-identical replicas.  Buffer
  // layout (floats): [0, 256) int words -- arrival word of (net, source rank) at 64 net + rank, sticky time-out word
  // at 240 --, then slabs [2 parities][3 networks][dp_world][XS].  p2p_seq0: optimiser steps exchanged through these
  // buffers before this launch (the same on every rank); arrival words only ever grow, so nothing is reset between
  // launches and a fast rank may write step k + 1 while a slow one still reads step k (other parity).
  float* p2p_peer[16];
  int p2p_rank;
  unsigned p2p_seq0;
  long long p2p_timeout;  // wall-clock ticks (100 MHz) a workgroup waits for its peers before it flags a time-out
  // 0: every exchange store carries the system-scope bits (sc0 sc1: written through to the memory side) and the
  // hand-off waits for the stores' acknowledgements only; 1: ordinary stores + LLVM's system-scope release / acquire
  // fences (an L2 write-back and invalidate per step: A/B switch OSA_P2P_FENCE=system)
  int p2p_fence;
};
#define OSA_P2P_HDR 256      // floats in front of the slabs
#define OSA_P2P_STICKY 240   // int index of the sticky time-out word

// Phase clocks (s_memtime deltas per phase, tools/phase_clocks.py, tools/dp_timing.py) are a COMPILE-TIME
// option: the runtime-checked version put a branch -- a scheduling barrier -- between all phases of the
// step.  Build with OSA_EXTRA_CFLAGS=-DOSA_PASS_CLOCKS to get them.
#ifdef OSA_PASS_CLOCKS
#define PTICK(k)                                      \
  do {                                                \
    if (a.dbg && tid == 0) {                          \
      const long long now_ = clock64();               \
      dbg_acc[k] += now_ - dbg_last;                  \
      dbg_last = now_;                                \
    }                                                 \
  } while (0)
#else
#define PTICK(k) do { } while (0)
#endif

// a * sa + b * sb with BOTH products rounded before the addition (no contraction into a fused multiply-add):
// the value is then symmetric in (a, sa) <-> (b, sb), which is what lets two peers that see each other's operand
// as "b" arrive at the same bits
__device__ __forceinline__ f32x4 osa_sym_sum(f32x4 a, float sa, f32x4 b, float sb) {
#pragma clang fp contract(off)
  const f32x4 pa = a * sa;
  const f32x4 pb = b * sb;
  return pa + pb;
}
__device__ __forceinline__ float osa_sym_sum1(float a, float sa, float b, float sb) {
#pragma clang fp contract(off)
  const float pa = a * sa;
  const float pb = b * sb;
  return pa + pb;
}

// Exchange stores of the peer-exchange pass: global stores with sc0 sc1 (system scope: the data is written through to
// the memory side -- the peer's HBM over xGMI, or this device's uncached buffer -- instead of waiting in this XCC's L2
// for a write-back).  Their completion is what s_waitcnt vmcnt(0) waits for before the arrival word goes out.
__device__ __forceinline__ void osa_store_sys(f32x4* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void osa_store_sys(float* p, float v) {
  asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}

// floats of dynamic LDS without the transposed W2 copy, and whether that copy still fits the 160 KB of a CU
__host__ __device__ constexpr int osa_pass_lds_floats(int KB, int OT) {
  return 64 * (16 * KB + PSPAD) + 64 * PSLD + 16 * OT * PSLD + 2 * 64 + 2 * 16 * OT + 4 * 64 * PSLD + 16 * KB * PSLD +
         2 * 16 * OT * PSLD + 64;
}
#ifndef OSA_PASS_W2T
#define OSA_PASS_W2T 1
#endif
__host__ __device__ constexpr bool osa_pass_has_w2t(int KB, int OT) {
  return OSA_PASS_W2T != 0 && (osa_pass_lds_floats(KB, OT) + 64 * PSLD) * 4 <= 160 * 1024;
}

// DPS: the gradient-only slab modes (osa_ppo_dp_step's data-parallel gradient step, the large-batch partial
// gradients) are their own instantiations: carrying them as a runtime branch cost the plain pass 1.3 % (same-box
// A/B: 9.22 -> 9.10 us per step; 72 slab-store addresses parked in AGPRs for the whole pass)
// SO ("small outputs"): EVERY network of the launch has 1-2 outputs (act_dim <= 2: SafetyPointGoal / CarGoal), so
// the VALU form of the output layer is taken at compile time: the run-time switch stood in front of every one of
// the 8 output-layer groups of the unrolled forward loop and in the backward / weight-gradient phases -- 16 scheduling
// barriers in the hottest code (same-box A/B: 9.11 -> 8.87 us per step)
// (Round 3's sliced data-parallel reduction, the two-stage large-batch pass and the per-network body split were
// measured slower and removed in round 4: profiles/HISTORY.md §7.)
template <int KB, int OT, bool MULTI, bool COOP, bool EXT, bool HIER, bool DPS, bool SO, bool P2P = false>
__device__ __forceinline__ void osa_ppo_pass_body(const OsaPassArgs& a, const int net, const int rk, const int pc0 = 0,
                                                  const int pn = 0) {
  static_assert(!P2P || (COOP && !HIER && !DPS && !EXT), "P2P is a form of the cooperative pass");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const OsaNet& nd = a.nd;
  constexpr bool coop = COOP;
#ifdef OSA_BODY_PART_ONLY
  // part_grad_kernel.hip: this translation unit only ever runs the balanced partial-gradient mode, so the mode
  // switches are compile-time constants there (no Adam-moment registers, no strided-chunk arithmetic)
  constexpr bool dp = true, part = true, contig = true;
#else
  const bool dp = DPS && a.dp_slabs != nullptr && !coop;
  const bool part = DPS && MULTI && a.part_stride != 0;  // (only the multi-chunk instantiations have this mode)
  const bool contig = part && a.part_stride < 0;         // balanced mode: chunks pc0 .. pc0 + pn - 1 of the minibatch
#endif
  const bool chunked = COOP && a.dp_chunk != 0;
  // chunk mode under data parallelism: peer rk = rank * cw + chunk (cw chunk workgroups per rank)
  // (HIER: its own instantiation, so that the plain chunk / data-parallel kernels keep their register allocation)
  const int nranks = (HIER && chunked && a.dp_ranks > 1) ? a.dp_ranks : 1;
  const int cw = chunked ? a.dp_world / nranks : 1;
  const int crank = chunked ? rk / cw : 0, cchunk = chunked ? rk - crank * cw : 0;
  const long roff = (part || P2P) ? 0 : (chunked ? (long)crank * a.M : (long)rk * a.M);  // (P2P: the rank's own arrays)
  const float* __restrict__ obs_p = a.obs + roff * a.ld_obs;
  const float* __restrict__ act_p = a.act + roff * a.ld_act;
  const float* __restrict__ logp_p = a.logp + roff;
  const float* __restrict__ advr_p = a.adv_r + roff;
  const float* __restrict__ advc_p = a.adv_c + roff;
  const long* __restrict__ perm_p = a.perm ? a.perm + roff : nullptr;
  constexpr int H = 64, HT = 4, OUTP = 16 * OT, INP = 16 * KB, W1LD = INP + PSPAD;
  // ---- LDS carve-up (all offsets multiples of 4 floats)
  float* sW1 = smem;                    // [H][W1LD]
  float* sW2 = sW1 + H * W1LD;          // [H][PSLD]
  float* sW3 = sW2 + H * PSLD;          // [OUTP][PSLD]
  float* sB1 = sW3 + OUTP * PSLD;       // [H]
  float* sB2 = sB1 + H;                 // [H]
  float* sB3 = sB2 + H;                 // [OUTP]
  float* sLS = sB3 + OUTP;              // [OUTP]
  float* sH1 = sLS + OUTP;              // [H][PSLD]   tiles: element (feature f, sample c)
  float* sH2 = sH1 + H * PSLD;
  float* sZ1 = sH2 + H * PSLD;
  float* sZ2 = sZ1 + H * PSLD;
  float* sX = sZ2 + H * PSLD;           // [INP][PSLD]
  float* sDO = sX + INP * PSLD;         // [OUTP][PSLD]
  float* sDL = sDO + OUTP * PSLD;       // [OUTP][PSLD]
  float* red = sDL + OUTP * PSLD;       // [4 waves][4] + spare
  // transposed copy of W2 (element (input feature, output feature) = W2[out][in]) for the backward pass: its A
  // operands W2^T[16t+i][k .. k+3] are then ONE 16-byte LDS read instead of four scalar ones (64 -> 16 LDS reads
  // per lane and step); kept in step by the Adam writers.  Only where the LDS budget allows (narrow observations).
  constexpr bool W2T = osa_pass_has_w2t(KB, OT);
  float* sW2T = red + 64;               // [H][PSLD]   (W2T only)

#ifdef OSA_BODY_OPAQUE_TID
  // (part_grad_kernel.hip calls this body inside a loop over a workgroup's segments: without the opaque copy LLVM
  // hoists every lane-derived address / mask computation out of that loop, where it stays live across the whole
  // body -- 443 -> 512 registers + 66 spills, 2.3 x the time per chunk)
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
#else
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
#endif
  const int i = j, cc = j;
  // 64-row chunks per (full) minibatch; compile-time 1 for B <= 64 (the reference's default batch)
  // (partial mode: this workgroup's chunks are rk, rk + stride, ... of the minibatch's ceil(B/64))
  const int nchunk_all = MULTI ? (a.B + 63) / 64 : 1;
  const bool strided = part;
  const int cstride = part ? (contig ? 1 : a.part_stride) : 1;
  const int cfirst = part ? (contig ? pc0 : rk) : (chunked ? cchunk : 0);
  const int nchunk = strided ? (contig ? pn : (nchunk_all - cfirst + cstride - 1) / cstride) : nchunk_all;
  auto pos_ok = [&](long cidx) -> bool {  // global chunk counter -> (minibatch, chunk)
    const long mb = cidx / nchunk, ch = cfirst + (cidx - mb * nchunk) * cstride;
    const long inb = ch * 64 + 16 * wave + j;
    return (mb < a.mb0 + a.nmb) && (inb < a.B) && (mb * a.B + inb < a.M);
  };
  auto row_of = [&](long cidx) -> long {  // raw (unconsumed) load of the permutation entry
    const long mb = cidx / nchunk, ch = cfirst + (cidx - mb * nchunk) * cstride;
    const long pc = pos_ok(cidx) ? mb * a.B + ch * 64 + 16 * wave + j : 0;
    return perm_p ? perm_p[pc] : pc;
  };
  // Round 6: the first two permutation entries are requested BEFORE the parameter loads -- the gather's first hop (a cold
  // trip to HBM for the index, then the rows) used to start only after the workgroup's own start-up and weight requests
  // (2 400 + 2 700 cycles per segment of the large-batch step by the part kernel's phase marks)
  const long cidx_first = (long)a.mb0 * nchunk;
  const long r0_first = row_of(cidx_first);
  const long r1_first = row_of(cidx_first + 1);
  // serial per-step chores (entropy, statistics) go to the first thread of wave 3: waves 0-2 also own the
  // bias-like parameters, so wave 3 is the one with slack before every barrier
  const bool leader = tid == 192;
  const int P = nd.P;
  float* __restrict__ gp = a.params + (long)net * P;
  float* __restrict__ gm = a.adam_m + (long)net * P;
  float* __restrict__ gv = a.adam_v + (long)net * P;
  const bool critic = net != 0;
  const bool is_actor = !critic;
  const int out_dim = critic ? 1 : nd.act_dim;
  // 1-2 real outputs (every critic; the actor of the 2-D action spaces): the output layer, its backward and
  // its weight gradient run on the VALU -- on gfx950 a float32 MFMA costs the same issue cycles as the
  // equivalent packed VALU math and does not overlap with it, so 48 MFMAs on a 16-wide tile that is 7/8
  // padding are pure waste (block-uniform switch; wider outputs keep the MFMA tiles)
  const bool small_out = SO || ((OT == 1) && out_dim <= 2);  // (SO: a compile-time `true`)

  // ---- load parameters into the LDS master copy (coalesced)
  // 16-byte loads (the blocks of a network are 64-byte aligned, rows multiples of 16 floats) and 16-byte LDS stores:
  // KB + 4 + OT loads per thread instead of 16 KB + 32 + 4 OT four-byte ones -- the prologue of a segment is paid
  // by every workgroup once per launch and twice by the two that straddle a network boundary.  Round 5: also in the
  // gradient-only instantiations (DPS), whose launches are ONE optimiser step each (osa_ppo_dp_step_phase: the per-step
  // all-reduce mode; `replicated-steps`) -- the prologue there is paid per step
#ifdef OSA_BODY_PART_ONLY
  constexpr bool VPRO = true;
#else
  constexpr bool VPRO = DPS;
#endif
  // Round 6 (VPRO): EVERY parameter load of the prologue is requested here, into registers, and nothing is done with the
  // values until the first minibatch's gather has been issued: an LDS store of a loaded bias right behind the W1
  // request made the workgroup sit out that round trip (3 200 cycles by the part kernel's phase marks), the gather's two
  // hops came after it (2 500) and W2 / W3 after those (900) -- three dependent trips per segment of the large-batch step
  f32x4 w1v_[VPRO ? KB : 1], w2v_[VPRO ? 4 : 1], w3v_[VPRO ? OT : 1];
  float b1v_ = 0.f, b2v_ = 0.f, b3v_ = 0.f, lsv_ = 0.f;
  if constexpr (VPRO) {
#pragma unroll
    for (int q = 0; q < KB; ++q) w1v_[q] = *reinterpret_cast<const f32x4*>(gp + nd.oW1 + 4 * (tid + 256 * q));
#pragma unroll
    for (int q = 0; q < 4; ++q) w2v_[q] = *reinterpret_cast<const f32x4*>(gp + nd.oW2 + 4 * (tid + 256 * q));
#pragma unroll
    for (int q = 0; q < OT; ++q) w3v_[q] = *reinterpret_cast<const f32x4*>(gp + nd.oW3 + 4 * (tid + 256 * q));
    if (tid < H) {
      b1v_ = gp[nd.ob1 + tid];
      b2v_ = gp[nd.ob2 + tid];
    }
    if (tid < OUTP) {
      b3v_ = gp[nd.ob3 + tid];
      lsv_ = gp[nd.oLS + tid];
    }
  } else {
    for (int e = tid; e < H * INP; e += 256) sW1[(e / INP) * W1LD + (e % INP)] = gp[nd.oW1 + e];
    if (tid < H) {
      sB1[tid] = gp[nd.ob1 + tid];
      sB2[tid] = gp[nd.ob2 + tid];
    }
    if (tid < OUTP) {
      sB3[tid] = gp[nd.ob3 + tid];
      sLS[tid] = gp[nd.oLS + tid];
    }
  }
  // (W2 / W3 are loaded after the first minibatch's gather has been issued, see below)
  OSA_PART_MARK(8);
  // ---- ownership: the parameters whose gradients this lane's accumulator tiles produce.
  //   W2[(16w+4g+r)][16ti+cc]  ti<4 | W1[(16w+4g+r)][16kb+cc] kb<KB | W3[(16o+4g+r)][16w+cc] o<OT
  //   + one bias-like scalar per thread: tid<64 b1 | <128 b2 | <128+OUTP b3 | <128+2*OUTP log_std
  f32x4 m2[HT], v2[HT], m1[KB], v1[KB], m3[OT], v3[OT];
  float mb_ = 0.f, vb_ = 0.f;
  int boff = -1;  // global offset of the owned bias-like scalar
  float* sbias = sB1;  // always a readable address (threads without a bias read it and discard)
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
      m2[ti][r] = dp ? 0.f : gm[off];
      v2[ti][r] = dp ? 0.f : gv[off];
    }
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int off = nd.oW1 + (16 * wave + 4 * g + r) * INP + 16 * kb + cc;
      m1[kb][r] = dp ? 0.f : gm[off];
      v1[kb][r] = dp ? 0.f : gv[off];
    }
#pragma unroll
  for (int o = 0; o < OT; ++o)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int off = nd.oW3 + (16 * o + 4 * g + r) * H + 16 * wave + cc;
      m3[o][r] = dp ? 0.f : gm[off];
      v3[o][r] = dp ? 0.f : gv[off];
    }
  if (boff >= 0 && !dp) {
    mb_ = gm[boff];
    vb_ = gv[boff];
  }
  OSA_PART_MARK(9);
  const int step0 = dp ? 0 : a.adam_step[net];
  const float lr = critic ? a.hp.lr_critic : a.hp.lr_actor;
  const float beta1 = a.hp.beta1, beta2 = a.hp.beta2, aeps = a.hp.adam_eps;
  // Adam's bias corrections of every step of the launch -- step_size = lr / (1 - beta1^t) and
  // 1 / sqrt(1 - beta2^t), float64 like torch -- are tabulated once (columns 10 + 2 net, 11 + 2 net of the
  // step's statistics row), 256 steps in parallel, instead of ~60 float64 VALU instructions (a division and
  // a square root) executed redundantly by every lane in every step.
  if (!dp) {
    for (int k = tid; k < a.nmb; k += 256) {
      const double t = (double)(step0 + k + 1);
      float* row = a.stats + (long)k * PNSTAT;
      row[10 + 2 * net] = (float)((double)lr / (1.0 - pow((double)beta1, t)));
      row[11 + 2 * net] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, t)));
    }
  }
  // terms that enter a minibatch's gradient ONCE (critics' L2 term, entropy bonus): partial sums get them later,
  // the chunks of a chunked pass from chunk 0
  const bool own_terms = !(chunked && cchunk != 0);
  const bool l2 = critic && a.hp.use_critic_norm && !part && own_terms;
  const float c2 = 2.f * a.hp.critic_norm_coef;
  float lam = 0.f;
  if (is_actor && a.lagrange) lam = *a.lagrange;
  const float inv_1p_lam = 1.f / (1.f + lam);
  // observation rows are 16-byte aligned with ld % 4 == 0 (checked by the entry points; the host pads
  // other layouts once per update): one code path, no branch inside the forward scheduling region
  constexpr bool vec_ok = true;
  const float* __restrict__ tgt = ((net == 1) ? a.tgt_r : a.tgt_c) + roff;

  // ---- prefetch machinery: everything this lane needs for its sample of one minibatch.
  // The loads are CONSUMER-FREE: addresses are clamped into the allocation instead of predicating the
  // loads, and all zero-masking (padding columns, invalid rows of a ragged last minibatch) happens
  // when the values are used one iteration later.  Otherwise every select on a loaded value makes the
  // compiler wait for the gather inside the prefetch (measured: ~4.5k stall cycles per step).
  struct Pre {
    f32x4 x[KB];
    float act[4 * OT];
    float logp, adv_r, adv_c, tgt;
    float old[EXT ? 4 * OT : 1];  // behaviour-policy mean of this lane's action dimensions (EXT)
    bool valid;
  };
  // The gather of the NEXT chunk lands in the SAME registers as the current one: the observation part is
  // re-issued as soon as layer 1 has consumed it, the per-sample scalars as soon as the loss has (no second
  // buffer: the kernel is register-bound -- 256 + ~250 AGPR in use -- and every spilled value costs VALU
  // moves that, on gfx950, add to the MFMA time instead of hiding under it).
  auto fetch_x = [&](long rr, Pre& q) {
    const float* xrow = obs_p + (int)rr * a.ld_obs;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int col0 = 16 * kb + 4 * g;
      if (vec_ok) {  // wave-uniform
        const int cl = (col0 + 4 <= a.ld_obs) ? col0 : a.ld_obs - 4;
        q.x[kb] = *reinterpret_cast<const f32x4*>(xrow + cl);
      } else {
        const int last = nd.obs_dim - 1;
        q.x[kb].x = xrow[min(col0, last)];
        q.x[kb].y = xrow[min(col0 + 1, last)];
        q.x[kb].z = xrow[min(col0 + 2, last)];
        q.x[kb].w = xrow[min(col0 + 3, last)];
      }
    }
  };
  auto fetch_s = [&](long rr, Pre& q) {
    // 32-bit index arithmetic (host guarantees M * ld < 2^31): 64-bit multiplies per load made the
    // prefetch issue itself cost ~1k cycles.  The actor-only loads sit behind a block-uniform branch.
    const int ri = (int)rr;
    if (is_actor) {
      const float* arow = act_p + ri * a.ld_act;
#pragma unroll
      for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int r = 0; r < 4; ++r) q.act[4 * o + r] = arow[min(16 * o + 4 * g + r, nd.act_dim - 1)];
      q.logp = logp_p[ri];
      q.adv_r = advr_p[ri];
      q.adv_c = advc_p[ri];
      q.tgt = 0.f;
      if constexpr (EXT) {
        const float* orow = a.old_mean + roff * a.ld_old_mean + ri * a.ld_old_mean;
#pragma unroll
        for (int o = 0; o < OT; ++o)
#pragma unroll
          for (int r = 0; r < 4; ++r) q.old[4 * o + r] = orow[min(16 * o + 4 * g + r, nd.act_dim - 1)];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4 * OT; ++k) q.act[k] = 0.f;
      q.logp = q.adv_r = q.adv_c = 0.f;
      q.tgt = tgt[ri];
    }
  };
  // column validity of this lane's 4-float chunks (static per lane)
  bool cm[KB][4];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) cm[kb][r] = (16 * kb + 4 * g + r) < nd.obs_dim;
  float dm[OT][4];  // 1 for this lane's real action dimensions, 0 for padding (static per lane)
#pragma unroll
  for (int o = 0; o < OT; ++o)
#pragma unroll
    for (int r = 0; r < 4; ++r) dm[o][r] = (16 * o + 4 * g + r) < nd.act_dim ? 1.f : 0.f;
  OSA_PART_MARK(10);
  Pre cur;
  long cidx = cidx_first;
  {
    const long r0 = r0_first;
    fetch_x(r0, cur);  // first gather in flight ...
    fetch_s(r0, cur);
    cur.valid = pos_ok(cidx);
  }
  long row_nxt = r1_first;
  OSA_PART_MARK(11);
  // ... while the remaining weights stream into LDS (matters for the one-step-per-launch dp mode)
  if constexpr (VPRO) {
    if (tid < H) {
      sB1[tid] = b1v_;
      sB2[tid] = b2v_;
    }
    if (tid < OUTP) {
      sB3[tid] = b3v_;
      sLS[tid] = lsv_;
    }
#pragma unroll
    for (int q = 0; q < KB; ++q) {  // float4 index e4 = tid + 256 q of W1[H][INP]: row e4 / (INP / 4), columns 4 (e4 % (INP / 4)) ..
      const int e4 = tid + 256 * q;
      *reinterpret_cast<f32x4*>(sW1 + (e4 / (INP / 4)) * W1LD + 4 * (e4 % (INP / 4))) = w1v_[q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e4 = tid + 256 * q, row = e4 >> 4, c4 = e4 & 15;
      *reinterpret_cast<f32x4*>(sW2 + row * PSLD + 4 * c4) = w2v_[q];
      if constexpr (W2T) {
#pragma unroll
        for (int k = 0; k < 4; ++k) sW2T[(4 * c4 + k) * PSLD + row] = w2v_[q][k];
      }
    }
#pragma unroll
    for (int q = 0; q < OT; ++q) {
      const int e4 = tid + 256 * q;
      *reinterpret_cast<f32x4*>(sW3 + (e4 >> 4) * PSLD + 4 * (e4 & 15)) = w3v_[q];
    }
  } else {
    for (int e = tid; e < H * H; e += 256) sW2[(e >> 6) * PSLD + (e & 63)] = gp[nd.oW2 + e];
    if constexpr (W2T)
      for (int e = tid; e < H * H; e += 256) sW2T[(e & 63) * PSLD + (e >> 6)] = gp[nd.oW2 + e];
    for (int e = tid; e < OUTP * H; e += 256) sW3[(e >> 6) * PSLD + (e & 63)] = gp[nd.oW3 + e];
  }
  OSA_PART_MARK(12);
  __syncthreads();  // LDS master copy complete
  OSA_PART_MARK(4);
#ifdef OSA_PASS_CLOCKS
  long long dbg_acc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long dbg_last = clock64();
#endif
  bool coop_dead = false;
  if constexpr (COOP) {
    if (a.dp_local) {
      // Verify the placement BEFORE anything is modified: every workgroup ORs its XCC bit into sync[4 + net]
      // and counts itself in at sync[7]; when all (active networks x dp_world) have arrived, every network's
      // mask must hold ONE bit.  Otherwise EVERY workgroup returns with parameters, Adam state and step counters
      // untouched and the sticky flag says why (2: a network on two XCCs; 1: somebody never arrived) -- the
      // caller repeats the pass with another placement (update.py).
      if (threadIdx.x == 0) {
        const int xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7;  // HW_REG_XCC_ID[3:0]
        __hip_atomic_fetch_or(a.dp_sync + 4 + net, 1 << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int expect = a.dp_world * __builtin_popcount(a.nets_mask & 7);
        // (release / acquire: the OR above is ordered before the arrival, the masks are read after the last one)
        int v = __hip_atomic_fetch_add(a.dp_sync + 7, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) + 1;
        int spins = 0, why = 0;
        while (v < expect) {
          __builtin_amdgcn_s_sleep(1);
          v = __hip_atomic_load(a.dp_sync + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (++spins > (1 << 21)) { why = 1; break; }
        }
        for (int n = 0; n < 3 && why == 0; ++n) {
          const int mask = __hip_atomic_load(a.dp_sync + 4 + n, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          if ((mask & (mask - 1)) != 0) why = 2;
        }
        if (why) __hip_atomic_store(a.dp_sync + 3, why, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // a late arriver sees a full count although an earlier one has already given up (and returned): the
        // sticky word decides for everybody, so nobody goes on to modify anything after a time-out
        if (!why) why = __hip_atomic_load(a.dp_sync + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        red[0] = (float)why;
      }
      __syncthreads();
      const bool bad_placement = red[0] != 0.f;
      __syncthreads();
      if (bad_placement) return;
    }
  }

  const float invB_full = 1.f / (float)a.B;
  for (int mb = a.mb0; mb < a.mb0 + a.nmb; ++mb) {
    const long mb_lo = (long)mb * a.B;
    const int Bcur = (int)(min(mb_lo + a.B, a.M) - mb_lo);
    float invB = invB_full;  // (an IEEE division per step otherwise: ~12 VALU instructions)
    if (Bcur != a.B) invB = 1.f / (float)Bcur;  // ragged last minibatch
    // weight-gradient accumulators of this optimiser step (summed over its 64-row chunks)
    f32x4 g2[HT], g1[KB], g3[OT];
#pragma unroll
    for (int ti = 0; ti < HT; ++ti) g2[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) g1[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < OT; ++o) g3[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // tabulated in the prologue, used by Adam; branch-free (the gradient-only modes, which may come without a
    // statistics buffer, read two floats of the parameter block instead and never use them)
    const float* __restrict__ bc_row = dp ? gp : a.stats + (long)(mb - a.mb0) * PNSTAT + 10 + 2 * net;
    const float step_size = bc_row[0], inv_bc2_sqrt = bc_row[1];
    float gb = 0.f, loss_part = 0.f, ratio_part = 0.f, ent_pre = 0.f;
    float cost_pen = 0.f;  // P3O penalty value of this step (EXT)
    if (is_actor && leader) {  // entropy of the pre-update policy (read before any Adam write)
      for (int d = 0; d < nd.act_dim; ++d) ent_pre += 1.41893853320467274178f + sLS[d];
      ent_pre /= (float)nd.act_dim;
    }
    for (int ch = 0; ch < nchunk; ++ch, ++cidx) {
    PTICK(0);
    const bool valid = cur.valid;
    const int c = 16 * wave + j;  // this lane's sample column in the [feature][sample] LDS tiles
    // Tile stores (S layout -> F layout for the weight-gradient contraction) are issued as soon as a
    // value is final, in the shadow of the MFMAs that follow, instead of in one block before barrier (A)
#define PUT_TILE(S, V, T)                                                           \
  _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) (S)[(16 * (T) + 4 * g + r_) * PSLD + c] = (V)[r_]
    // deferred masking of the prefetched observation chunks (padding columns)
    // only the last K block can contain padding columns; rows of a ragged last minibatch were gathered from
    // clamped (valid, finite) addresses and meet dL/dout = 0 downstream, so they need no masking
#pragma unroll
    for (int r = 0; r < 4; ++r) cur.x[KB - 1][r] = cm[KB - 1][r] ? cur.x[KB - 1][r] : 0.f;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) PUT_TILE(sX, cur.x[kb], kb);
    // ================= forward (S layout; weights from the LDS master) =================
    // Software-pipelined in two halves of the output tiles (A = tiles 0,1; B = tiles 2,3): the tanh of a
    // finished half is VALU work placed in the shadow of the other half's / the next layer's MFMAs
    // (a v_mfma_f32_16x16x4 occupies the matrix pipe for 32 cycles, the wave can issue ~6 VALU ops
    // meanwhile).  Two accumulators alternate, so no MFMA waits on its own predecessor.
    f32x4 h1[HT], h2[HT], out[OT];
#pragma unroll
    for (int t = 0; t < HT; ++t) h1[t] = *reinterpret_cast<const f32x4*>(sB1 + 16 * t + 4 * g);
#pragma unroll
    for (int t = 0; t < HT; ++t) h2[t] = *reinterpret_cast<const f32x4*>(sB2 + 16 * t + 4 * g);
#pragma unroll
    for (int o = 0; o < OT; ++o) out[o] = *reinterpret_cast<const f32x4*>(sB3 + 16 * o + 4 * g);
    // The three layers are one sequence of MFMA groups (8 MFMAs on two alternating accumulator tiles, or
    // 4*OT for the output layer); the A fragments of group k+1 are read from the LDS master BEFORE the
    // MFMAs of group k issue (double buffer; OSA_SB pins the order of the group's reads and MFMAs),
    // so no MFMA waits for an LDS round trip -- with one wave per SIMD nothing else would hide it.
    //   groups 0 .. 2KB-1        layer 1: tiles (T0, T0+1), T0 = 0 then 2, K block kb over the input
    //   groups 2KB .. 2KB+7      layer 2: (0,kb0) (0,kb1) (2,kb0) (2,kb1) (0,kb2) (0,kb3) (2,kb2) (2,kb3)
    //   groups 2KB+8 .. 2KB+11   output layer, K blocks 0..3
#define OSA_SB() __builtin_amdgcn_sched_barrier(0)  // (A/B: 0 beats 0x676 = "VALU/SALU/VMEM may cross" and none)
    constexpr int NG1 = 2 * KB, NG = NG1 + 8 + 4;
    auto load_group = [&](int gi, f32x4 (&dst)[2]) {
      if (gi < NG1) {
        const int T0 = 2 * (gi / KB), kb = gi % KB;
        dst[0] = *reinterpret_cast<const f32x4*>(sW1 + (16 * T0 + i) * W1LD + 16 * kb + 4 * g);
        dst[1] = *reinterpret_cast<const f32x4*>(sW1 + (16 * (T0 + 1) + i) * W1LD + 16 * kb + 4 * g);
      } else if (gi < NG1 + 8) {
        const int q = gi - NG1, T0 = 2 * ((q >> 1) & 1), kb = 2 * (q >> 2) + (q & 1);
        dst[0] = *reinterpret_cast<const f32x4*>(sW2 + (16 * T0 + i) * PSLD + 16 * kb + 4 * g);
        dst[1] = *reinterpret_cast<const f32x4*>(sW2 + (16 * (T0 + 1) + i) * PSLD + 16 * kb + 4 * g);
      } else if (!small_out) {
        const int kb = gi - NG1 - 8;
#pragma unroll
        for (int o = 0; o < OT; ++o)
          dst[o] = *reinterpret_cast<const f32x4*>(sW3 + (16 * o + i) * PSLD + 16 * kb + 4 * g);
      }
    };
    auto mm_group = [&](int gi, const f32x4 (&w)[2]) {
      if (gi < NG1) {
        const int T0 = 2 * (gi / KB), kb = gi % KB;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          h1[T0] = OSA_MFMA(w[0][s], cur.x[kb][s], h1[T0]);
          h1[T0 + 1] = OSA_MFMA(w[1][s], cur.x[kb][s], h1[T0 + 1]);
        }
      } else if (gi < NG1 + 8) {
        const int q = gi - NG1, T0 = 2 * ((q >> 1) & 1), kb = 2 * (q >> 2) + (q & 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          h2[T0] = OSA_MFMA(w[0][s], h1[kb][s], h2[T0]);
          h2[T0 + 1] = OSA_MFMA(w[1][s], h1[kb][s], h2[T0 + 1]);
        }
      } else if (!small_out) {
        const int kb = gi - NG1 - 8;
#pragma unroll
        for (int o = 0; o < OT; ++o)
#pragma unroll
          for (int s = 0; s < 4; ++s) out[o] = OSA_MFMA(w[o][s], h2[kb][s], out[o]);
      }
    };
    f32x4 wq[2][2];
    load_group(0, wq[0]);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi + 1 < NG) load_group(gi + 1, wq[(gi + 1) & 1]);
      OSA_SB();
      mm_group(gi, wq[gi & 1]);
      OSA_SB();
      // ---- VALU / memory work placed in the shadow of the MFMA groups that follow
      if (gi == KB) {  // first group of tiles 2,3 issued; tiles 0,1 of layer 1 are complete
        h1[0] = osa_tanh4(h1[0]);
        h1[1] = osa_tanh4(h1[1]);
      }
      if (gi == NG1 - 1) fetch_x(row_nxt, cur);  // layer 1 has consumed x: next chunk's rows, in place
      if (gi == NG1 + 1) {  // layer 2 on K blocks 0,1 under way
        PUT_TILE(sH1, h1[0], 0);
        PUT_TILE(sH1, h1[1], 1);
        h1[2] = osa_tanh4(h1[2]);
        h1[3] = osa_tanh4(h1[3]);
      }
      if (gi == NG1 + 6) {  // tiles 0,1 of layer 2 complete (groups +4, +5)
        PUT_TILE(sH1, h1[2], 2);
        PUT_TILE(sH1, h1[3], 3);
        h2[0] = osa_tanh4(h2[0]);
        h2[1] = osa_tanh4(h2[1]);
      }
      if (gi == NG1 + 8) {  // output layer on K blocks 0,1 under way; tiles 2,3 of layer 2 complete
        PUT_TILE(sH2, h2[0], 0);
        PUT_TILE(sH2, h2[1], 1);
        h2[2] = osa_tanh4(h2[2]);
        h2[3] = osa_tanh4(h2[3]);
      }
    }
    PUT_TILE(sH2, h2[2], 2);
    PUT_TILE(sH2, h2[3], 3);
    if (small_out) {
      // out[d] = b3[d] + sum_f W3[d][f] h2[f]: each lane holds 16 of the 64 features of its sample
      float p0 = 0.f, p1 = 0.f;
      f32x4 w0[HT], w1[HT];
#pragma unroll
      for (int t = 0; t < HT; ++t) {  // all eight reads in flight before the dot products
        w0[t] = *reinterpret_cast<const f32x4*>(sW3 + 16 * t + 4 * g);
        w1[t] = *reinterpret_cast<const f32x4*>(sW3 + PSLD + 16 * t + 4 * g);  // zero row if 1 output
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < HT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p0 = fmaf(w0[t][r], h2[t][r], p0);
          p1 = fmaf(w1[t][r], h2[t][r], p1);
        }
      }
      p0 = osa_sum_over_groups(p0);
      p1 = osa_sum_over_groups(p1);
      if (g == 0) {  // lane group 0 holds output dimensions 0..3 of its sample (out[] starts as the bias)
        out[0][0] += p0;
        out[0][1] += p1;
      }
    }

    PTICK(1);
    // ================= loss, dL/d(out) =================
    f32x4 dO[OT], dLS[OT];
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      dO[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dLS[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (is_actor) {
      // Branch-free over the action dimensions (dm = 1 for this lane's real dimensions, 0 for padding): the
      // divergent `if (d < act_dim && valid)` form cost exec-mask juggling on the critical (actor) block.
      float lp = 0.f;
      f32x4 zv[OT], ivar[OT];
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        // Normal.log_prob with sigma = exp(log_std): 1/var = exp(-2 log_std) (one hardware exp2),
        // log(sigma) = log_std (the reference takes log(exp(log_std)): equal to float32 rounding)
        const f32x4 ls = *reinterpret_cast<const f32x4*>(sLS + 16 * o + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float dmr = dm[o][r];
          const float iv = __builtin_amdgcn_exp2f(ls[r] * -2.88539008177792681472f) * dmr;
          const float z = cur.act[4 * o + r] - out[o][r];
          zv[o][r] = z;
          ivar[o][r] = iv;
          lp += -0.5f * (z * z) * iv - ls[r] * dmr - 0.91893853320467274178f * dmr;
        }
      }
      lp = osa_sum_over_groups(lp);
      // ---- EXT: per-sample KL(pi_theta || pi_old) (torch.distributions.kl._kl_normal_normal), FOCOPS'
      // trust mask with the reference's broadcast semantics (focops.py:84-88: the surrogate term sees the
      // minibatch MEAN of the mask), P3O's kappa * relu(mean(ratio * A_c) + excess) -- see osa_mb_grad_kernel
      float kl = 0.f, mask = 1.f, mask_mean = 1.f, cost_w = 0.f;
      f32x4 dkl_mu[OT], dkl_ls[OT];
      if constexpr (EXT) {
#pragma unroll
        for (int o = 0; o < OT; ++o) {
          const f32x4 ls = *reinterpret_cast<const f32x4*>(sLS + 16 * o + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // hardware exp2 only: sigma/sigma0 = exp(ls - ls0), log(var_ratio) = 2 (ls - ls0), 1/sigma0 = exp(-ls0)
            const int d = min(16 * o + 4 * g + r, nd.act_dim - 1);
            const float ls0 = a.old_log_std[d], dl = ls[r] - ls0;
            const float q = __builtin_amdgcn_exp2f(dl * 1.44269504088896340736f), var_ratio = q * q;
            const float isd0 = __builtin_amdgcn_exp2f(ls0 * -1.44269504088896340736f);
            const float dmu = out[o][r] - cur.old[4 * o + r];
            const float u = dmu * isd0;
            kl += 0.5f * (var_ratio + u * u - 1.f - 2.f * dl) * dm[o][r];
            dkl_mu[o][r] = u * isd0 * dm[o][r];
            dkl_ls[o][r] = (var_ratio - 1.f) * dm[o][r];
          }
        }
        kl = osa_sum_over_groups(kl);
      }
      const float ratio = valid ? __builtin_amdgcn_exp2f((lp - cur.logp) * 1.44269504088896340736f) : 0.f;
      if constexpr (EXT) {
        if (a.ext_mask_eta >= 0.f || a.ext_cost_kappa > 0.f) {  // block-uniform; single-chunk minibatches
          mask = (a.ext_mask_eta < 0.f || (valid && kl <= a.ext_mask_eta)) ? 1.f : 0.f;
          float pm = (valid && g == 0) ? mask : 0.f;
          float pc = (valid && g == 0) ? ratio * cur.adv_c : 0.f;
          pm = osa_wave_sum_dpp(pm);
          pc = osa_wave_sum_dpp(pc);
          __syncthreads();  // `red` is free (its previous readers passed barrier C)
          if (lane == 0) {
            red[4 * wave + 0] = pm;
            red[4 * wave + 1] = pc;
          }
          __syncthreads();
          const float tm = red[0] + red[4] + red[8] + red[12], tc = red[1] + red[5] + red[9] + red[13];
          if (a.ext_mask_eta >= 0.f) mask_mean = tm * invB;
          if (a.ext_cost_kappa > 0.f) {
            const float pen = tc * invB + a.ext_cost_excess;
            if (pen > 0.f) cost_w = a.ext_cost_kappa;
            cost_pen = a.ext_cost_kappa * fmaxf(pen, 0.f);
          }
          __syncthreads();  // `red` is reused by the norm reduction
        }
      }
      {
        const float adv = (cur.adv_r - lam * cur.adv_c) * inv_1p_lam;
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
        float dklw = 0.f;
        if constexpr (EXT) {
          const float rs = a.ext_ratio_scale * mask_mean;
          li = li * rs + a.ext_kl_coef * kl * mask;
          dratio = dratio * rs + cost_w * cur.adv_c;
          dklw = valid ? a.ext_kl_coef * mask * invB : 0.f;
        }
        const float dlogp = valid ? dratio * ratio * invB : 0.f;
        if (g == 0 && valid) {
          loss_part += li;
          ratio_part += ratio;
        }
#pragma unroll
        for (int o = 0; o < OT; ++o) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float z = zv[o][r], iv = ivar[o][r];
            dO[o][r] = dlogp * z * iv;
            dLS[o][r] = dlogp * (z * z * iv - dm[o][r]);
            if constexpr (EXT) {
              dO[o][r] += dklw * dkl_mu[o][r];
              dLS[o][r] += dklw * dkl_ls[o][r];
            }
          }
        }
      }
    } else if (valid) {
      const float diff = out[0][0] - cur.tgt;
      if (g == 0) {
        loss_part += diff * diff;
        dO[0][0] = 2.f * diff * invB;
      }
    }

    PTICK(2);
    fetch_s(row_nxt, cur);  // the loss has consumed the per-sample scalars: next chunk's, in place
    cur.valid = pos_ok(cidx + 1);
    row_nxt = row_of(cidx + 2);
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      PUT_TILE(sDO, dO[o], o);
      PUT_TILE(sDL, dLS[o], o);
    }
    // ================= backward through the hidden layers =================
    // The transposed weight fragments are scalar LDS reads (A[i][k] = W^T[16t+i][k]); with one wave per SIMD
    // nothing hides their latency, so every block of reads is issued BEFORE the MFMAs of the previous block
    // (double buffer + sched_barrier: the compiler otherwise sinks each read next to its consumer and
    // waits ~100 cycles in front of every MFMA pair).
    f32x4 z2[HT], z1[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) z2[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float wt[2][4][HT];
    // A fragments of the W2^T product, K block kb: wt[.][s][t] = W2[16 kb + 4 g + s][16 t + i]
    auto load_w2t = [&](int kb, float (&dst)[4][HT]) {
      if constexpr (W2T) {
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(sW2T + (16 * t + i) * PSLD + 16 * kb + 4 * g);
#pragma unroll
          for (int s = 0; s < 4; ++s) dst[s][t] = v[s];
        }
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < HT; ++t) dst[s][t] = sW2[(16 * kb + 4 * g + s) * PSLD + 16 * t + i];
      }
    };
    if (small_out) {
      // z2[f] = sum_d W3[d][f] dO[d] with dO of this lane's sample held by lane group 0 (d = r)
      const float d0 = __shfl(dO[0][0], j, 64), d1 = __shfl(dO[0][1], j, 64);
      load_w2t(0, wt[0]);  // first W2^T block, in flight meanwhile
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(sW3 + 16 * t + 4 * g);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(sW3 + PSLD + 16 * t + 4 * g);
        z2[t] = w0 * d0 + w1 * d1;
      }
    } else {
      float w3t[OT][4][HT];
#pragma unroll
      for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < HT; ++t) w3t[o][s][t] = sW3[(16 * o + 4 * g + s) * PSLD + 16 * t + i];
      load_w2t(0, wt[0]);  // first W2^T block, in flight under the W3^T MFMAs
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int s = 0; s < 4; ++s)  // the 4 tiles t are independent accumulators
#pragma unroll
          for (int t = 0; t < HT; ++t) z2[t] = OSA_MFMA(w3t[o][s][t], dO[o][s], z2[t]);
    }
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      z2[t] = z2[t] * (1.f - h2[t] * h2[t]);
      PUT_TILE(sZ2, z2[t], t);
    }
#pragma unroll
    for (int t = 0; t < HT; ++t) z1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < HT; ++kb) {
      if (kb + 1 < HT) load_w2t(kb + 1, wt[(kb + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < HT; ++t) z1[t] = OSA_MFMA(wt[kb & 1][s][t], z2[kb][s], z1[t]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      z1[t] = z1[t] * (1.f - h1[t] * h1[t]);
      PUT_TILE(sZ1, z1[t], t);
    }
#undef PUT_TILE
    PTICK(3);
    __syncthreads();  // (A) tiles complete
    PTICK(4);
    // ================= weight gradients (registers) =================
    {
      f32x4 a2[4], a1[4];
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        a2[sb] = *reinterpret_cast<const f32x4*>(sZ2 + (16 * wave + i) * PSLD + 16 * sb + 4 * g);
        a1[sb] = *reinterpret_cast<const f32x4*>(sZ1 + (16 * wave + i) * PSLD + 16 * sb + 4 * g);
      }
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        f32x4 b[HT];
#pragma unroll
        for (int ti = 0; ti < HT; ++ti)
          b[ti] = *reinterpret_cast<const f32x4*>(sH1 + (16 * ti + i) * PSLD + 16 * sb + 4 * g);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int ti = 0; ti < HT; ++ti) g2[ti] = OSA_MFMA(a2[sb][s], b[ti][s], g2[ti]);
      }
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        f32x4 b[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
          b[kb] = *reinterpret_cast<const f32x4*>(sX + (16 * kb + i) * PSLD + 16 * sb + 4 * g);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) g1[kb] = OSA_MFMA(a1[sb][s], b[kb][s], g1[kb]);
      }
      if (small_out) {
        // dW3[d][f] = sum_s dO[d][s] h2[f][s] for this lane's column f = 16 wave + cc: lane group g takes
        // samples 16g .. 16g+15, the four partial sums are combined across the groups
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f32x4 hv = *reinterpret_cast<const f32x4*>(sH2 + (16 * wave + i) * PSLD + 16 * g + 4 * k);
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(sDO + 16 * g + 4 * k);
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(sDO + PSLD + 16 * g + 4 * k);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            q0 = fmaf(a0[r], hv[r], q0);
            q1 = fmaf(a1[r], hv[r], q1);
          }
        }
        q0 = osa_sum_over_groups(q0);
        q1 = osa_sum_over_groups(q1);
        if (g == 0) {
          g3[0][0] += q0;
          g3[0][1] += q1;
        }
      } else {
#pragma unroll
        for (int o = 0; o < OT; ++o) {
#pragma unroll
          for (int sb = 0; sb < 4; ++sb) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(sDO + (16 * o + i) * PSLD + 16 * sb + 4 * g);
            const f32x4 b = *reinterpret_cast<const f32x4*>(sH2 + (16 * wave + i) * PSLD + 16 * sb + 4 * g);
            g3[o] = OSA_MFMA(av.x, b.x, g3[o]);
            g3[o] = OSA_MFMA(av.y, b.y, g3[o]);
            g3[o] = OSA_MFMA(av.z, b.z, g3[o]);
            g3[o] = OSA_MFMA(av.w, b.w, g3[o]);
          }
        }
      }
    }
    PTICK(5);
    // bias-like gradient owned by this thread: row sum over the 64 samples of the chunk.  Branch-free
    // (threads without a bias read row 0 and discard) with all 16 reads in flight before the adds.
    {
      const float* srow = (tid < H) ? sZ1 + tid * PSLD
                          : (tid < 2 * H) ? sZ2 + (tid - H) * PSLD
                          : (tid < 2 * H + OUTP) ? sDO + (tid - 2 * H) * PSLD
                          : (tid < 2 * H + 2 * OUTP) ? sDL + (tid - 2 * H - OUTP) * PSLD
                                                     : sZ1;
      f32x4 q[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) q[k] = *reinterpret_cast<const f32x4*>(srow + 4 * k);
      __builtin_amdgcn_sched_barrier(0);
      float rs = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        rs += q[k].x;
        rs += q[k].y;
        rs += q[k].z;
        rs += q[k].w;
      }
      gb = (boff >= 0) ? ((MULTI && ch > 0) ? gb + rs : rs) : 0.f;
    }
    if (ch + 1 < nchunk) __syncthreads();  // tiles free for the next chunk of this step
    if (ch == 0) OSA_PART_MARK(5);
    }  // chunks
    OSA_PART_MARK(6);
    // (not in the partial-gradient mode: osa_slab_reduce_finalize_kernel adds the entropy term ONCE to the sum of
    // the slabs -- until round 3 every partial slab carried it as well, i.e. the large-batch step applied it
    // 1 + workgroups times whenever entropy_coef != 0 (the YAML default is 0; found by
    // test_large_batch_pass_equals_per_step_launches))
    if (boff >= 0 && is_actor && boff >= nd.oLS && (boff - nd.oLS) < nd.act_dim && a.hp.entropy_coef != 0.f && own_terms &&
        !part)
      gb -= a.hp.entropy_coef / (float)nd.act_dim;
    // ================= + 2*coef*w (critics), squared norms (packed f32 math) =================
    // this lane's parameters: all LDS reads issued up front (one latency for the lot), kept in registers
    // until the Adam update below
    f32x4 w2r[HT], w1r[KB], w3r[OT];
#pragma unroll
    for (int ti = 0; ti < HT; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) w2r[ti][r] = sW2[(16 * wave + 4 * g + r) * PSLD + 16 * ti + cc];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) w1r[kb][r] = sW1[(16 * wave + 4 * g + r) * W1LD + 16 * kb + cc];
#pragma unroll
    for (int o = 0; o < OT; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r) w3r[o][r] = sW3[(16 * o + 4 * g + r) * PSLD + 16 * wave + cc];
    float wb = *sbias;
    wb = (boff >= 0) ? wb : 0.f;
    __builtin_amdgcn_sched_barrier(0);
    // cooperative mode: this rank's UNSCALED gradient tiles are published as soon as they are final -- the
    // stores drain to memory while the norm is reduced; the clip factor follows in the slab's tail and every
    // peer forms the same products g_r * clip_r
    constexpr int NT = HT + KB + OT, XS = NT * 1024 + 256 + PNSTAT;
    float* __restrict__ xbase = nullptr;
    f32x4* __restrict__ xs4 = nullptr;
    // (P2P: offset of this step's slab set of this network inside EVERY rank's exchange buffer)
    const long p2p_off = P2P ? (long)OSA_P2P_HDR + (((long)((a.p2p_seq0 + (unsigned)(mb - a.mb0)) & 1u) * 3 + net) * a.dp_world) * XS : 0;
    if constexpr (P2P) {
      xbase = a.p2p_peer[a.p2p_rank] + p2p_off;  // the rank's own buffer: where the sum is formed
      xs4 = reinterpret_cast<f32x4*>(xbase + (long)rk * XS);
    } else if constexpr (coop) {
      // (chunk mode: the slabs of THIS rank's chunk group; own slab = chunk index)
      xbase = a.dp_slabs + (((long)((mb - a.mb0) & 1) * 3 + net) * a.dp_world + (chunked ? crank * cw : 0)) * XS;
      xs4 = reinterpret_cast<f32x4*>(xbase + (long)(chunked ? cchunk : rk) * XS);
    }
    f32x4 acc_g = {0.f, 0.f, 0.f, 0.f}, acc_p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ti = 0; ti < HT; ++ti) {
      const f32x4 w = w2r[ti];
      if (l2) g2[ti] = g2[ti] + w * c2;
      if constexpr (coop && !P2P) xs4[ti * 256 + tid] = g2[ti];  // (P2P: the own gradient stays in registers)
      acc_p = acc_p + w * w;
      acc_g = acc_g + g2[ti] * g2[ti];
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const f32x4 w = w1r[kb];
      if (l2) g1[kb] = g1[kb] + w * c2;
      if constexpr (coop && !P2P) xs4[(HT + kb) * 256 + tid] = g1[kb];
      acc_p = acc_p + w * w;
      acc_g = acc_g + g1[kb] * g1[kb];
    }
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      const f32x4 w = w3r[o];
      if (l2) g3[o] = g3[o] + w * c2;
      if constexpr (coop && !P2P) xs4[(HT + KB + o) * 256 + tid] = g3[o];
      acc_p = acc_p + w * w;
      acc_g = acc_g + g3[o] * g3[o];
    }
    float gsq = (acc_g.x + acc_g.y) + (acc_g.z + acc_g.w);
    float psq = (acc_p.x + acc_p.y) + (acc_p.z + acc_p.w);
    if (boff >= 0) {
      if (l2) gb += c2 * wb;
      psq += wb * wb;
      gsq += gb * gb;
    }
    if constexpr (coop && !P2P) reinterpret_cast<float*>(xs4)[NT * 1024 + tid] = (boff >= 0) ? gb : 0.f;
    if constexpr (P2P) {
      // the same tiles into every OTHER rank's buffer (posted writes: over xGMI where the peer is another device);
      // they drain while the norm is reduced, the clip factor and the arrival word follow below
      for (int q = 1; q < a.dp_world; ++q) {
        int pr = a.p2p_rank + q;
        pr = pr >= a.dp_world ? pr - a.dp_world : pr;  // (every rank starts with its right-hand neighbour: the 7 links in parallel)
        f32x4* __restrict__ d4 = reinterpret_cast<f32x4*>(a.p2p_peer[pr] + p2p_off + (long)rk * XS);
#pragma unroll
        for (int ti = 0; ti < HT; ++ti) osa_store_sys(d4 + ti * 256 + tid, g2[ti]);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) osa_store_sys(d4 + (HT + kb) * 256 + tid, g1[kb]);
#pragma unroll
        for (int o = 0; o < OT; ++o) osa_store_sys(d4 + (HT + KB + o) * 256 + tid, g3[o]);
        osa_store_sys(reinterpret_cast<float*>(d4) + NT * 1024 + tid, (boff >= 0) ? gb : 0.f);
      }
    }
    // ---- block reduction of (gsq, psq, loss, ratio): wave shuffles, then a fixed-order sum
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
    PTICK(6);
    __syncthreads();  // (B)
    PTICK(7);
    const float t_gsq = red[0] + red[4] + red[8] + red[12];
    const float t_psq = red[1] + red[5] + red[9] + red[13];
    const float t_loss = red[2] + red[6] + red[10] + red[14];
    const float t_ratio = red[3] + red[7] + red[11] + red[15];
    const float total_norm = sqrtf(t_gsq);
    float coef = 1.f;
    if (a.hp.use_max_grad_norm) {
      coef = a.hp.max_grad_norm / (total_norm + 1e-6f);
      coef = coef > 1.f ? 1.f : coef;
    }
    if (dp) {
      // publish the locally clipped gradient of (net, rk) and its statistics; reduce + Adam follow in
      // osa_dp_apply_kernel (clip-then-average order of policy_gradient.py:437-442)
      float* __restrict__ slab = a.dp_slabs + ((long)net * a.dp_world + rk) * (P + PNSTAT);
      const float gs = (a.hp.use_max_grad_norm && !part) ? coef : 1.f;
#pragma unroll
      for (int ti = 0; ti < HT; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[nd.oW2 + (16 * wave + 4 * g + r) * H + 16 * ti + cc] = g2[ti][r] * gs;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[nd.oW1 + (16 * wave + 4 * g + r) * INP + 16 * kb + cc] = g1[kb][r] * gs;
#pragma unroll
      for (int o = 0; o < OT; ++o)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[nd.oW3 + (16 * o + 4 * g + r) * H + 16 * wave + cc] = g3[o][r] * gs;
      if (boff >= 0) slab[boff] = gb * gs;
      if (critic && tid < OUTP) slab[nd.oLS + tid] = 0.f;
      if (leader && part) {  // raw sums: osa_slab_reduce_kernel adds the slabs and normalises
        slab[P + 0] = t_loss;
        slab[P + 1] = t_ratio;
      } else if (leader) {
        slab[P + 0] = t_loss * invB - (is_actor ? a.hp.entropy_coef * ent_pre : 0.f);
        slab[P + 1] = t_ratio * invB;
        slab[P + 2] = t_psq;
        slab[P + 3] = total_norm;
        slab[P + 4] = ent_pre;
      }
      return;  // nmb == 1 in this mode: nothing else to do (weights, moments untouched)
    }
    float st_loss = t_loss * invB, st_ratio = t_ratio * invB, st_psq = t_psq, st_norm = total_norm,
          st_ent = ent_pre;
    if (is_actor && own_terms) st_loss -= a.hp.entropy_coef * ent_pre;
    bool apply_clip = a.hp.use_max_grad_norm != 0;
    if constexpr (coop) {
      // ---- the gradient tiles are on their way (exchange layout: one f32x4 per thread per tile, 1 KB
      // contiguous per wave instruction); complete the slab with the clip factor and the statistics
      const int W = chunked ? cw : a.dp_world;  // peers of this hand-off
      const float gs = (apply_clip && !chunked) ? coef : 1.f;  // (chunk mode: the SUM is clipped, below)
      if constexpr (P2P) {
        // the slab's tail (statistics, clip factor) into every rank's buffer: lane q of the leader's wave serves rank q
        if (wave == 3) {  // (wave-uniform; the entropy terms live in the leader = lane 0 of this wave)
          const float l_loss = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(st_loss)));
          const float l_ent = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(st_ent)));
          if (lane < a.dp_world) {
            float* tq = a.p2p_peer[lane] + p2p_off + (long)rk * XS + NT * 1024 + 256;
            osa_store_sys(reinterpret_cast<f32x4*>(tq), (f32x4){l_loss, st_ratio, st_psq, st_norm});
            osa_store_sys(tq + 4, l_ent);
            osa_store_sys(tq + 5, gs);
          }
        }
      } else if (leader) {
        float* t = reinterpret_cast<float*>(xs4) + NT * 1024 + 256;
        t[0] = st_loss; t[1] = st_ratio; t[2] = st_psq; t[3] = st_norm; t[4] = st_ent; t[5] = gs;
      }
      if constexpr (P2P) {
        // ---- one-shot exchange: every thread's slab stores are performed at SYSTEM scope (release), then thread q
        // stores the step's sequence number into this rank's arrival word in rank q's buffer and polls the word rank
        // q stores into OUR buffer -- dp_world words polled in parallel by dp_world lanes of wave 0
        if (a.p2p_fence) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        } else {
          // (the stores were written through at system scope: their acknowledgements are all that is waited for)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_s_waitcnt(0);
        }
        __syncthreads();
        PTICK(12);
        const unsigned target = a.p2p_seq0 + (unsigned)(mb - a.mb0) + 1u;
        if (tid < a.dp_world) {
          unsigned* theirs = reinterpret_cast<unsigned*>(a.p2p_peer[tid]) + 64 * net + a.p2p_rank;
          __hip_atomic_store(theirs, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          unsigned* own = reinterpret_cast<unsigned*>(a.p2p_peer[a.p2p_rank]);
          unsigned* mine = own + 64 * net + tid;
          if (!coop_dead) {
            long long t0 = 0;
            int spins = 0;
            while ((int)(__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - target) < 0) {
              __builtin_amdgcn_s_sleep(1);
              if ((++spins & 1023) == 0) {  // the wall clock only every ~1k polls
                const long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                if (now - t0 > a.p2p_timeout ||
                    __hip_atomic_load(own + OSA_P2P_STICKY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
                  // a peer is gone (or another workgroup of this rank gave up): flag it, never hang the GPU
                  __hip_atomic_store(own + OSA_P2P_STICKY, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                  break;
                }
              }
            }
          }
        }
        __syncthreads();
        if (!coop_dead) coop_dead = __hip_atomic_load(reinterpret_cast<unsigned*>(a.p2p_peer[a.p2p_rank]) + OSA_P2P_STICKY,
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
        // the own buffer is uncached (never in this XCC's L2): the vector L1 is what has to forget its lines
        if (a.p2p_fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      } else {
      if (a.dp_uncached || a.dp_local) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        // the slab stores have been performed: at device scope (uncached memory) / in the XCC's L2 (local)
        __builtin_amdgcn_s_waitcnt(0);
      } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      }
      __syncthreads();
      PTICK(12);
      if (tid == 0) {
        int* cnt = (nranks > 1) ? a.dp_sync + 8 + 16 * net + crank : a.dp_sync + net;
        const int target = W * (mb - a.mb0 + 1);
        // relaxed atomics: ordering comes from the agent-scope fences on either side of the barriers
        // (a release/acquire atomic would write back / invalidate the L2 a second time)
        int seen = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        if (!coop_dead) {
          int spins = 0;
          while (seen < target) {
            __builtin_amdgcn_s_sleep(1);
            seen = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (++spins > (1 << 21)) {  // peers not co-resident / lost: flag it, never hang the GPU
              __hip_atomic_store(a.dp_sync + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              coop_dead = true;
              break;
            }
          }
        }
      }
      __syncthreads();
      // always an L1 invalidate: the vector L1 also keeps lines of UNCACHED memory between two reads of the
      // same address (observed in wide_split_kernel.hip with small working sets; the double-buffered 38 KB
      // slabs here never showed it, which is luck, not a guarantee)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      PTICK(10);
      // ---- sum the W gradients in rank order (same order on every peer), average
      f32x4 s2[HT], s1[KB], s3[OT];
      float sb_ = 0.f;
#pragma unroll
      for (int ti = 0; ti < HT; ++ti) s2[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) s1[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int o = 0; o < OT; ++o) s3[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (W == 2) {
        // two peers (two chunks of a 128-row minibatch, or two ranks): own gradient from the registers + the
        // peer's slab.  Both sides form the same two products (g * clip factor; 1 in chunk mode) and a two-operand
        // float sum is commutative, so they get the same bits without walking the slabs in rank order.
        const float* __restrict__ xr = xbase + (long)((chunked ? cchunk : rk) ^ 1) * XS;
        const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(xr);
        f32x4 t[NT];
#pragma unroll
        for (int q = 0; q < NT; ++q) t[q] = x4[q * 256 + tid];
        const float tb = xr[NT * 1024 + tid], tg = xr[NT * 1024 + 256 + 5];
        // (both products ROUNDED, then added: a fused multiply-add would round g_own * gs differently from the
        // peer's view of the same product, and the two replicas would drift apart by an ulp)
#pragma unroll
        for (int ti = 0; ti < HT; ++ti) s2[ti] = osa_sym_sum(g2[ti], gs, t[ti], tg);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) s1[kb] = osa_sym_sum(g1[kb], gs, t[HT + kb], tg);
#pragma unroll
        for (int o = 0; o < OT; ++o) s3[o] = osa_sym_sum(g3[o], gs, t[HT + KB + o], tg);
        sb_ = osa_sym_sum1((boff >= 0) ? gb : 0.f, gs, tb, tg);
      } else {
      // RU ranks per trip: their loads are all in flight together (one memory round trip per trip,
      // not per rank); the clamped duplicate loads of a ragged last trip are simply not added
      constexpr int RU = 4;
      for (int r0 = 0; r0 < W; r0 += RU) {
        f32x4 t[RU][NT];
        float tb[RU], tg[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          int src = min(r0 + u, W - 1);
          // P2P: the own gradient never went to memory -- its slot of the rank-ordered sum is filled from the
          // registers below; the load is pointed at a neighbour's slab (lines the trip requests anyway)
          if (P2P && src == rk) src = (src > 0) ? src - 1 : (W > 1 ? 1 : 0);
          const float* __restrict__ xr = xbase + (long)src * XS;
          const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(xr);
#pragma unroll
          for (int q = 0; q < NT; ++q) t[u][q] = x4[q * 256 + tid];
          tb[u] = xr[NT * 1024 + tid];
          tg[u] = xr[NT * 1024 + 256 + 5];
        }
        if constexpr (P2P) {
#pragma unroll
          for (int u = 0; u < RU; ++u) {
            if (r0 + u == rk) {  // workgroup-uniform: the same bits the peers read from this rank's slab
#pragma unroll
              for (int ti = 0; ti < HT; ++ti) t[u][ti] = g2[ti];
#pragma unroll
              for (int kb = 0; kb < KB; ++kb) t[u][HT + kb] = g1[kb];
#pragma unroll
              for (int o = 0; o < OT; ++o) t[u][HT + KB + o] = g3[o];
              tb[u] = (boff >= 0) ? gb : 0.f;
              tg[u] = gs;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          if (r0 + u < W) {  // workgroup-uniform
#pragma unroll
            for (int ti = 0; ti < HT; ++ti) s2[ti] = s2[ti] + t[u][ti] * tg[u];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) s1[kb] = s1[kb] + t[u][HT + kb] * tg[u];
#pragma unroll
            for (int o = 0; o < OT; ++o) s3[o] = s3[o] + t[u][HT + KB + o] * tg[u];
            sb_ += tb[u] * tg[u];
          }
        }
      }
      }
      const float invW = chunked ? 1.f : 1.f / (float)W;
#pragma unroll
      for (int ti = 0; ti < HT; ++ti) g2[ti] = s2[ti] * invW;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) g1[kb] = s1[kb] * invW;
#pragma unroll
      for (int o = 0; o < OT; ++o) g3[o] = s3[o] * invW;
      gb = sb_ * invW;
      PTICK(11);
      apply_clip = false;  // already clipped per rank (clip-then-average, policy_gradient.py:437-442)
      float chunk_norm = 0.f;
      if (chunked) {
        // the minibatch's gradient is the SUM of its chunks' (every chunk scaled by 1 / rows of the minibatch): its
        // norm decides the clip factor, as in a single-process step over all B rows
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ti = 0; ti < HT; ++ti) acc2 = acc2 + g2[ti] * g2[ti];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) acc2 = acc2 + g1[kb] * g1[kb];
#pragma unroll
        for (int o = 0; o < OT; ++o) acc2 = acc2 + g3[o] * g3[o];
        float q2 = (acc2.x + acc2.y) + (acc2.z + acc2.w);
        if (boff >= 0) q2 += gb * gb;
        q2 = osa_wave_sum_dpp(q2);
        if (lane == 0) red[wave] = q2;
        __syncthreads();
        chunk_norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
        if (a.hp.use_max_grad_norm) {
          coef = a.hp.max_grad_norm / (chunk_norm + 1e-6f);
          coef = coef > 1.f ? 1.f : coef;
          apply_clip = true;
        }
      }
      if (leader && (P2P || (chunked ? cchunk == 0 : rk == 0))) {  // what Logger.get_stats averages across ranks
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < W; ++r) {
          const float* t = xbase + (long)r * XS + NT * 1024 + 256;
          for (int k = 0; k < 5; ++k) acc[k] += t[k];
        }
        if (chunked) {  // loss and ratio: sums of the chunks' shares; parameter norm and entropy: chunk 0's
          const float* t0 = xbase + NT * 1024 + 256;
          st_loss = acc[0]; st_ratio = acc[1]; st_psq = t0[2]; st_norm = chunk_norm; st_ent = t0[4];
        } else {
          st_loss = acc[0] * invW; st_ratio = acc[1] * invW; st_psq = acc[2] * invW;
          st_norm = acc[3] * invW; st_ent = acc[4] * invW;
        }
      }
      if constexpr (HIER) {
        // ---- second hand-off (chunk mode under data parallelism): the rank sums, across ranks.  Every chunk
        // workgroup of a rank holds the same sum: chunk c publishes the tiles q = c, c + cw, ... (chunk 0 also
        // the bias-like row, the rank's clip factor and its statistics); ALL peers arrive; everybody adds the
        // nranks rank slabs x clip factor in rank order.
        float* x2 = a.dp_slabs + (long)2 * 3 * a.dp_world * XS + (((long)((mb - a.mb0) & 1) * 3 + net) * nranks) * XS;
        f32x4* o4 = reinterpret_cast<f32x4*>(x2 + (long)crank * XS);
#pragma unroll
        for (int ti = 0; ti < HT; ++ti)
          if (ti % cw == cchunk) o4[ti * 256 + tid] = g2[ti];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
          if ((HT + kb) % cw == cchunk) o4[(HT + kb) * 256 + tid] = g1[kb];
#pragma unroll
        for (int o = 0; o < OT; ++o)
          if ((HT + KB + o) % cw == cchunk) o4[(HT + KB + o) * 256 + tid] = g3[o];
        if (cchunk == 0) {
          float* row = x2 + (long)crank * XS + NT * 1024;
          row[tid] = (boff >= 0) ? gb : 0.f;
          if (leader) {
            float* t = row + 256;
            t[0] = st_loss; t[1] = st_ratio; t[2] = st_psq; t[3] = st_norm; t[4] = st_ent;
            t[5] = apply_clip ? coef : 1.f;
          }
        }
        if (a.dp_uncached || a.dp_local) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_s_waitcnt(0);
        } else {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        }
        __syncthreads();
        if (tid == 0) {
          int* cnt = a.dp_sync + net;
          const int target = a.dp_world * (mb - a.mb0 + 1);
          int seen = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
          if (!coop_dead) {
            int spins = 0;
            while (seen < target) {
              __builtin_amdgcn_s_sleep(1);
              seen = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (++spins > (1 << 21)) {
                __hip_atomic_store(a.dp_sync + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                coop_dead = true;
                break;
              }
            }
          }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
        for (int ti = 0; ti < HT; ++ti) s2[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) s1[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < OT; ++o) s3[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
        sb_ = 0.f;
        constexpr int RU2 = 4;
        for (int r0 = 0; r0 < nranks; r0 += RU2) {
          f32x4 t[RU2][NT];
          float tb[RU2], tg[RU2];
#pragma unroll
          for (int u = 0; u < RU2; ++u) {
            const float* __restrict__ xr = x2 + (long)min(r0 + u, nranks - 1) * XS;
            const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(xr);
#pragma unroll
            for (int q = 0; q < NT; ++q) t[u][q] = x4[q * 256 + tid];
            tb[u] = xr[NT * 1024 + tid];
            tg[u] = xr[NT * 1024 + 256 + 5];
          }
#pragma unroll
          for (int u = 0; u < RU2; ++u) {
            if (r0 + u < nranks) {  // workgroup-uniform
#pragma unroll
              for (int ti = 0; ti < HT; ++ti) s2[ti] = s2[ti] + t[u][ti] * tg[u];
#pragma unroll
              for (int kb = 0; kb < KB; ++kb) s1[kb] = s1[kb] + t[u][HT + kb] * tg[u];
#pragma unroll
              for (int o = 0; o < OT; ++o) s3[o] = s3[o] + t[u][HT + KB + o] * tg[u];
              sb_ += tb[u] * tg[u];
            }
          }
        }
        const float invR = 1.f / (float)nranks;
#pragma unroll
        for (int ti = 0; ti < HT; ++ti) g2[ti] = s2[ti] * invR;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) g1[kb] = s1[kb] * invR;
#pragma unroll
        for (int o = 0; o < OT; ++o) g3[o] = s3[o] * invR;
        gb = sb_ * invR;
        apply_clip = false;  // clipped per rank above (clip-then-average)
        if (leader && rk == 0) {
          float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
          for (int r = 0; r < nranks; ++r) {
            const float* t = x2 + (long)r * XS + NT * 1024 + 256;
            for (int k = 0; k < 5; ++k) acc[k] += t[k];
          }
          st_loss = acc[0] * invR; st_ratio = acc[1] * invR; st_psq = acc[2] * invR;
          st_norm = acc[3] * invR; st_ent = acc[4] * invR;
        }
      }
    }
    // ================= Adam on the owned parameters; LDS master updated in place =================
    const float gscale = apply_clip ? coef : 1.f;
    auto put_w2 = [&](int ti, f32x4 w) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sW2[(16 * wave + 4 * g + r) * PSLD + 16 * ti + cc] = w[r];
      if constexpr (W2T) *reinterpret_cast<f32x4*>(sW2T + (16 * ti + cc) * PSLD + 16 * wave + 4 * g) = w;
    };
    auto put_w1 = [&](int kb, f32x4 w) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sW1[(16 * wave + 4 * g + r) * W1LD + 16 * kb + cc] = w[r];
    };
    auto put_w3 = [&](int o, f32x4 w) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sW3[(16 * o + 4 * g + r) * PSLD + 16 * wave + cc] = w[r];
    };
#pragma unroll
    for (int ti = 0; ti < HT; ++ti) {
      {
        f32x4 w = w2r[ti];
        w = osa_adam_update4(g2[ti] * gscale, m2[ti], v2[ti], w, beta1, beta2, step_size, inv_bc2_sqrt, aeps);
        put_w2(ti, w);
      }
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      {
        f32x4 w = w1r[kb];
        w = osa_adam_update4(g1[kb] * gscale, m1[kb], v1[kb], w, beta1, beta2, step_size, inv_bc2_sqrt, aeps);
        put_w1(kb, w);
      }
    }
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      {
        f32x4 w = w3r[o];
        w = osa_adam_update4(g3[o] * gscale, m3[o], v3[o], w, beta1, beta2, step_size, inv_bc2_sqrt, aeps);
        put_w3(o, w);
      }
    }
    {
      float nb = wb;
      if (boff >= 0) {
        float mv_ = mb_, vv_ = vb_;
        nb = osa_adam_update(gb * gscale, mv_, vv_, wb, beta1, beta2, step_size, inv_bc2_sqrt, aeps);
        *sbias = nb;
        mb_ = mv_;
        vb_ = vv_;
      }
    }
    PTICK(8);
    // ---- statistics of this optimiser step
    if (leader && (P2P || rk == 0)) {
      float* st = a.stats + (long)(mb - a.mb0) * PNSTAT;
      if (is_actor) {
        st[2] = st_loss;
        st[3] = st_ratio;
        st[4] = st_ent;
        st[7] = st_norm;
        if constexpr (EXT) {
          if (a.ext_cost_kappa > 0.f) st[10] = cost_pen;  // (this row's bias-correction entry is consumed)
        }
      } else {
        st[net - 1] = st_loss;
        st[4 + net] = st_psq;
        st[7 + net] = st_norm;
      }
    }
    __syncthreads();  // (C) master copy updated, tiles and `red` free for the next minibatch
    PTICK(9);
  }
#ifdef OSA_PASS_CLOCKS
  if (a.dbg && tid == 0 && (P2P || rk == 0))
    for (int k = 0; k < 13; ++k) a.dbg[net * 16 + k] = dbg_acc[k];
#endif
  if (!P2P && rk != 0) return;  // cooperative mode: the peers' copies are identical, rank 0's is written back
  // ---- write back parameters and Adam state
  for (int e = tid; e < H * INP; e += 256) gp[nd.oW1 + e] = sW1[(e / INP) * W1LD + (e % INP)];
  for (int e = tid; e < H * H; e += 256) gp[nd.oW2 + e] = sW2[(e >> 6) * PSLD + (e & 63)];
  for (int e = tid; e < OUTP * H; e += 256) gp[nd.oW3 + e] = sW3[(e >> 6) * PSLD + (e & 63)];
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
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int off = nd.oW1 + (16 * wave + 4 * g + r) * INP + 16 * kb + cc;
        gm[off] = m1[kb][r];
        gv[off] = v1[kb][r];
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
  (void)out_dim;
}

