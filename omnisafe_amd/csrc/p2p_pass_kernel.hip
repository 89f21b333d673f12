// ONE-SHOT PEER EXCHANGE for the data-parallel update (round 6): the persistent pass of ppo_pass_body.h with real ranks.
//
// Replaces, per optimiser step, the reference's `clip_grad_norm_` -> `distributed.avg_grads` (19 blocking all-reduces:
// omnisafe/algorithms/on_policy/base/policy_gradient.py:437-443, 478-483, 519-524; omnisafe/utils/distributed.py:167-198)
// -> `optimizer.step()`.  SURVEY.md 8(e) "xGMI mapping": the messages are <= 336 KB, i.e. latency-bound -- so no ring,
// no RCCL on the step path at all: every rank runs the SINGLE-GPU persistent pass on its own 64 rows (3 workgroups, one
// per network, weights in LDS, Adam moments in registers) and after the local clip each workgroup writes its gradient
// slab straight into the exchange buffer of every rank (memory mapped through hipIpcOpenMemHandle: posted writes over
// the 7 xGMI links in parallel, one hop), stores the step's sequence number into its arrival word there, polls its own
// buffer's arrival words, sums the slabs IN RANK ORDER (identical arithmetic => bit-identical replicas, no parameter
// traffic) and applies Adam.  What the replicated design pays per GPU for W ranks -- W times the gradient work of a
// step -- becomes W - 1 posted slab writes.
//
// The exchange buffers are UNCACHED device memory (hipExtMallocWithFlags(hipDeviceMallocUncached)): a peer's writes
// arrive at the memory side and must not meet stale lines in the owner's per-XCC L2.
#include <stdlib.h>
#include <string.h>

#include "ppo_pass_body.h"

long long* osa_pass_dbg_ptr();  // ppo_pass_kernel.hip (phase clocks: -DOSA_PASS_CLOCKS builds)

template <int KB, int OT, bool MULTI, bool SO>
__global__ __launch_bounds__(256, 1) void osa_ppo_p2p_pass_kernel(OsaPassArgs a) {
  // blocks 0, 8, 16 of a 17-block grid: the three networks on ONE XCC (the rows all three gather come from HBM once)
  if (blockIdx.x & 7) return;
  const int net = blockIdx.x >> 3;
  if (!((a.nets_mask >> net) & 1)) return;
  osa_ppo_pass_body<KB, OT, MULTI, true, false, false, false, SO, true>(a, net, a.p2p_rank);
}

static size_t osa_p2p_lds_bytes(int KB, int OT) {
  const size_t fl = (size_t)osa_pass_lds_floats(KB, OT) + (osa_pass_has_w2t(KB, OT) ? 64 * PSLD : 0);
  return fl * sizeof(float);
}

template <int KB, int OT, bool MULTI, bool SO>
static int osa_launch_p2p(const OsaPassArgs& a, hipStream_t stream) {
  static OsaPerDeviceOnce attr_set;
  const size_t lds = osa_p2p_lds_bytes(KB, OT);
  if (lds > 160 * 1024) return OSA_EUNSUPPORTED;
  if (attr_set.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&osa_ppo_p2p_pass_kernel<KB, OT, MULTI, SO>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return OSA_EHIP;
    attr_set.set();
  }
  hipLaunchKernelGGL((osa_ppo_p2p_pass_kernel<KB, OT, MULTI, SO>), dim3(17), dim3(256), lds, stream, a);
  return hipGetLastError() == hipSuccess ? OSA_OK : OSA_EHIP;
}

template <int KB, int OT, bool MULTI>
static int osa_launch_p2p_so(const OsaPassArgs& a, hipStream_t stream) {
  if constexpr (OT == 1) {
    if (a.nd.act_dim <= 2) return osa_launch_p2p<KB, OT, MULTI, true>(a, stream);
  }
  return osa_launch_p2p<KB, OT, MULTI, false>(a, stream);
}

// exchange buffers of this process: own allocations (base, bytes) and peers' buffers opened through IPC handles
static const int OSA_MAX_P2P = 256;
static char* g_p2p_base[OSA_MAX_P2P];
static size_t g_p2p_bytes[OSA_MAX_P2P];  // 0: a peer's buffer (opened, not owned)

static int osa_p2p_slot(const void* p) {
  for (int k = 0; k < OSA_MAX_P2P; ++k)
    if (g_p2p_base[k] == static_cast<const char*>(p)) return k;
  return -1;
}

extern "C" {

size_t osa_p2p_exchange_floats(int obs_dim, int act_dim, int hidden, int world) {
  if (!osa_ppo_pass_supported(obs_dim, act_dim, hidden) || world < 1 || world > 16) return 0;
  const OsaNet nd = osa_make_net(obs_dim, act_dim, hidden);
  const size_t xs = (size_t)(4 + nd.KB + nd.OUTP / 16) * 1024 + 256 + PNSTAT;
  return (size_t)OSA_P2P_HDR + (size_t)2 * 3 * world * xs;
}

int osa_p2p_exchange_alloc(size_t floats, float** out, void* ipc_handle64) {
  OSA_REQUIRE(out != nullptr && ipc_handle64 != nullptr && floats >= OSA_P2P_HDR);
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C ABI hands IPC handles over as 64 bytes");
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, floats * sizeof(float), hipDeviceMallocUncached) != hipSuccess || !p) {
    (void)hipGetLastError();
    return OSA_EUNSUPPORTED;
  }
  if (hipMemset(p, 0, floats * sizeof(float)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(p);
    return OSA_EHIP;
  }
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, p) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(p);
    return OSA_EUNSUPPORTED;
  }
  for (int k = 0; k < OSA_MAX_P2P; ++k)
    if (!g_p2p_base[k]) {
      g_p2p_base[k] = static_cast<char*>(p);
      g_p2p_bytes[k] = floats * sizeof(float);
      memcpy(ipc_handle64, &h, 64);
      *out = static_cast<float*>(p);
      return OSA_OK;
    }
  (void)hipFree(p);
  return OSA_EUNSUPPORTED;
}

int osa_p2p_exchange_open(const void* ipc_handle64, float** out) {
  OSA_REQUIRE(out != nullptr && ipc_handle64 != nullptr);
  hipIpcMemHandle_t h;
  memcpy(&h, ipc_handle64, 64);
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !p) {
    (void)hipGetLastError();
    return OSA_EUNSUPPORTED;
  }
  for (int k = 0; k < OSA_MAX_P2P; ++k)
    if (!g_p2p_base[k]) {
      g_p2p_base[k] = static_cast<char*>(p);
      g_p2p_bytes[k] = 0;
      *out = static_cast<float*>(p);
      return OSA_OK;
    }
  (void)hipIpcCloseMemHandle(p);
  return OSA_EUNSUPPORTED;
}

int osa_p2p_exchange_release(float* p) {
  const int k = osa_p2p_slot(p);
  if (k < 0) return OSA_EINVAL;
  const bool own = g_p2p_bytes[k] != 0;
  g_p2p_base[k] = nullptr;
  g_p2p_bytes[k] = 0;
  const hipError_t e = own ? hipFree(p) : hipIpcCloseMemHandle(p);
  if (e != hipSuccess) (void)hipGetLastError();
  return e == hipSuccess ? OSA_OK : OSA_EHIP;
}

int osa_p2p_exchange_timed_out(const float* own, int* flag) {
  OSA_REQUIRE(own != nullptr && flag != nullptr);
  const int k = osa_p2p_slot(own);
  if (k < 0 || g_p2p_bytes[k] == 0) return OSA_EINVAL;
  unsigned v = 0;
  if (hipMemcpy(&v, reinterpret_cast<const unsigned*>(own) + OSA_P2P_STICKY, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess)
    return OSA_EHIP;
  *flag = (int)v;
  return OSA_OK;
}

int osa_ppo_p2p_pass(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                     int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                     const float* logp, const float* target_value_r, const float* target_value_c,
                     const float* adv_r, const float* adv_c, const long* perm, long M, int B, int world, int rank,
                     float* const* peers, unsigned seq0, double timeout_s, const float* lagrange,
                     const osa_ppo_hparams* hp, int loss_kind, int nets_mask, float* step_stats, void* stream) {
  if (!osa_ppo_pass_supported(obs_dim, act_dim, hidden)) return OSA_EUNSUPPORTED;
  if (world < 1 || world > 16 || B > 64 * 32) return OSA_EUNSUPPORTED;
  OSA_REQUIRE(params && adam_m && adam_v && adam_step && obs && act && logp && hp && step_stats && peers);
  OSA_REQUIRE(target_value_r && target_value_c && adv_r && adv_c && M > 0 && B > 0 && rank >= 0 && rank < world);
  OSA_REQUIRE(ld_obs >= obs_dim && ld_act >= act_dim && timeout_s > 0.0);
  if ((double)M * ld_obs >= 2147483647.0 || (double)M * ld_act >= 2147483647.0) return OSA_EUNSUPPORTED;
  if (ld_obs % 4 != 0 || (reinterpret_cast<uintptr_t>(obs) & 15) != 0) return OSA_EUNSUPPORTED;  // pad rows
  OsaPassArgs a = {};
  for (int q = 0; q < world; ++q) {
    OSA_REQUIRE(peers[q] != nullptr && osa_p2p_slot(peers[q]) >= 0);  // only buffers this library allocated / opened
    a.p2p_peer[q] = peers[q];
  }
  {  // the rank's own buffer is its own allocation and large enough
    const int k = osa_p2p_slot(peers[rank]);
    OSA_REQUIRE(g_p2p_bytes[k] >= osa_p2p_exchange_floats(obs_dim, act_dim, hidden, world) * sizeof(float));
  }
  a.p2p_rank = rank; a.p2p_seq0 = seq0; a.p2p_timeout = (long long)(timeout_s * 1e8);
  {  // OSA_P2P_FENCE=system: LLVM's system-scope fences instead of written-through stores (A/B switch)
    static const bool sysf = getenv("OSA_P2P_FENCE") && !strcmp(getenv("OSA_P2P_FENCE"), "system");
    a.p2p_fence = sysf ? 1 : 0;
  }
  a.ext_ratio_scale = 1.f; a.ext_mask_eta = -1.f;
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
  a.dbg = osa_pass_dbg_ptr(); a.dp_slabs = nullptr; a.dp_world = world; a.mb0 = 0; a.dp_sync = nullptr; a.part_stride = 0;
  a.dp_uncached = 1; a.dp_local = 0; a.dp_chunk = 0; a.dp_ranks = 1; a.one_xcc = 1;
  const int KB = a.nd.KB, OT = a.nd.OUTP / 16;
  hipStream_t st = osa_stream(stream);
#define OSA_P2P_CASE(K, O) \
  if (KB == K && OT == O) return (B > 64) ? osa_launch_p2p_so<K, O, true>(a, st) : osa_launch_p2p_so<K, O, false>(a, st)
  OSA_P2P_CASE(1, 1); OSA_P2P_CASE(2, 1); OSA_P2P_CASE(3, 1); OSA_P2P_CASE(4, 1); OSA_P2P_CASE(5, 1);
  OSA_P2P_CASE(6, 1); OSA_P2P_CASE(1, 2); OSA_P2P_CASE(2, 2); OSA_P2P_CASE(3, 2); OSA_P2P_CASE(4, 2);
  OSA_P2P_CASE(5, 2); OSA_P2P_CASE(6, 2);
#undef OSA_P2P_CASE
  return OSA_EUNSUPPORTED;
}

}  // extern "C"
