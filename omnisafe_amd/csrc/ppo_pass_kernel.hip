// Launch wrappers and C-ABI entry points of the persistent pass kernels; the kernel body (one template, all
// modes) lives in ppo_pass_body.h, shared with part_grad_kernel.hip (balanced partial gradients, round 4).
#include "ppo_pass_body.h"

template <int KB, int OT, bool MULTI, bool COOP, bool EXT, bool HIER = false, bool DPS = false, bool SO = false>
__global__ __launch_bounds__(256, 1) void osa_ppo_pass_kernel(OsaPassArgs a) {
  int net_ = blockIdx.x, rk_ = blockIdx.y;  // rk: virtual rank (0 outside the data-parallel mode)
  if constexpr (!COOP) {
    if (a.one_xcc) {
      if (blockIdx.x & 7) return;
      net_ = blockIdx.x >> 3;
    }
  }
  if constexpr (COOP) {
    if (a.dp_local == 1) {  // one XCC per network: 8 x world blocks, blocks 3..7 (mod 8) have nothing to do
      // (dp_local == 3: test hook -- the one-XCC protocol on the 3 x world grid, so that the placement check trips)
      net_ = blockIdx.x & 7;
      rk_ = blockIdx.x >> 3;
      if (net_ >= 3) return;
    }
  }
  const int net = net_, rk = rk_;
  if (!((a.nets_mask >> net) & 1)) return;
  osa_ppo_pass_body<KB, OT, MULTI, COOP, EXT, HIER, DPS, SO>(a, net, rk);
}

// ------------------------------------------------------------------------------------------------
static long long* g_osa_pass_dbg = nullptr;

// (library-internal: the phase-clock buffer for the other translation units that instantiate the pass body)
long long* osa_pass_dbg_ptr() { return g_osa_pass_dbg; }

static size_t osa_pass_lds_bytes(int KB, int OT) {
  const size_t fl = (size_t)osa_pass_lds_floats(KB, OT) + (osa_pass_has_w2t(KB, OT) ? 64 * PSLD : 0);
  return fl * sizeof(float);
}

template <int KB, int OT, bool MULTI, bool COOP = false, bool EXT = false, bool HIER = false, bool DPS = false,
          bool SO = false>
static int osa_launch_pass(const OsaPassArgs& a, hipStream_t stream, int grid_y = 1) {
  const dim3 grid = (COOP && a.dp_local == 1) ? dim3(8 * grid_y) : ((!COOP && a.one_xcc) ? dim3(17) : dim3(3, grid_y));
  static OsaPerDeviceOnce attr_set;
  const size_t lds = osa_pass_lds_bytes(KB, OT);
  if (lds > 160 * 1024) return OSA_EUNSUPPORTED;
  if (attr_set.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&osa_ppo_pass_kernel<KB, OT, MULTI, COOP, EXT, HIER, DPS, SO>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return OSA_EHIP;
    attr_set.set();
  }
  if constexpr (COOP) {
    // the 3 x world workgroups meet at an arrival counter every step, so they MUST be co-resident: a
    // cooperative launch makes the runtime verify that (occupancy x CUs >= grid) instead of inferring it
    // from the CU count, and refuses the launch otherwise
    // (OSA_DP_PLAIN_LAUNCH=1: A/B switch of tools/dp_timing.py -- plain launch behind the occupancy check)
    static bool coop_refused = getenv("OSA_DP_PLAIN_LAUNCH") != nullptr && getenv("OSA_DP_PLAIN_LAUNCH")[0] == '1';
    if (!coop_refused) {
      OsaPassArgs arg = a;
      void* kargs[] = {&arg};
      const hipError_t e = hipLaunchCooperativeKernel(
          reinterpret_cast<const void*>(&osa_ppo_pass_kernel<KB, OT, MULTI, COOP, EXT, HIER, DPS, SO>), grid,
          dim3(256), kargs, (unsigned)lds, stream);
      if (e == hipSuccess) return OSA_OK;
      (void)hipGetLastError();
      if (e == hipErrorCooperativeLaunchTooLarge) return OSA_EUNSUPPORTED;  // caller takes the stepwise path
      coop_refused = true;
    }
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &per_cu, reinterpret_cast<const void*>(&osa_ppo_pass_kernel<KB, OT, MULTI, COOP, EXT, HIER, DPS, SO>), 256, lds) !=
            hipSuccess ||
        hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return OSA_EHIP;
    if ((long)per_cu * cus < (long)grid.x * grid.y) return OSA_EUNSUPPORTED;
  }
  hipLaunchKernelGGL((osa_ppo_pass_kernel<KB, OT, MULTI, COOP, EXT, HIER, DPS, SO>), grid, dim3(256), lds, stream, a);
  return hipGetLastError() == hipSuccess ? OSA_OK : OSA_EHIP;
}

// the SO instantiation where it applies (one output tile and act_dim <= 2: every network then has 1-2 outputs)
template <int KB, int OT, bool MULTI, bool COOP = false, bool EXT = false, bool HIER = false>
static int osa_launch_pass_so(const OsaPassArgs& a, hipStream_t stream, int grid_y = 1) {
  if constexpr (OT == 1) {
    if (a.nd.act_dim <= 2) return osa_launch_pass<KB, OT, MULTI, COOP, EXT, HIER, false, true>(a, stream, grid_y);
  }
  return osa_launch_pass<KB, OT, MULTI, COOP, EXT, HIER>(a, stream, grid_y);
}

extern "C" {

int osa_debug_set_pass_clock_buffer(long long* dev_ptr) {
  g_osa_pass_dbg = dev_ptr;
  return OSA_OK;
}

int osa_ppo_pass_supported(int obs_dim, int act_dim, int hidden) {
  if (hidden != 64 || obs_dim < 1 || act_dim < 1 || act_dim > 32) return 0;
  const int KB = (obs_dim + 15) / 16, OT = (act_dim + 15) / 16;
  return (KB <= 6 && OT <= 2 && osa_pass_lds_bytes(KB, OT) <= 160 * 1024) ? 1 : 0;
}

int osa_ppo_pass(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                 int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                 const float* logp, const float* target_value_r, const float* target_value_c,
                 const float* adv_r, const float* adv_c, const long* perm, long M, int B,
                 const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                 float* step_stats, void* stream) {
  return osa_ppo_pass_ext(obs_dim, act_dim, hidden, params, adam_m, adam_v, adam_step, obs, ld_obs, act, ld_act,
                          logp, target_value_r, target_value_c, adv_r, adv_c, perm, M, B, lagrange, hp,
                          loss_kind, nets_mask, step_stats, nullptr, stream);
}

int osa_ppo_pass_ext(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                 int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                 const float* logp, const float* target_value_r, const float* target_value_c,
                 const float* adv_r, const float* adv_c, const long* perm, long M, int B,
                 const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                 float* step_stats, const osa_surrogate_ext* ext, void* stream) {
  if (!osa_ppo_pass_supported(obs_dim, act_dim, hidden)) return OSA_EUNSUPPORTED;
  OSA_REQUIRE(params && adam_m && adam_v && adam_step && obs && act && logp && hp && step_stats);
  OSA_REQUIRE(target_value_r && target_value_c && adv_r && adv_c && M > 0 && B > 0);
  OSA_REQUIRE(ld_obs >= obs_dim && ld_act >= act_dim);
  if ((double)M * ld_obs >= 2147483647.0 || (double)M * ld_act >= 2147483647.0) return OSA_EUNSUPPORTED;
  if (ld_obs % 4 != 0 || (reinterpret_cast<uintptr_t>(obs) & 15) != 0) return OSA_EUNSUPPORTED;  // pad rows
  OsaPassArgs a = {};
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
  a.dbg = g_osa_pass_dbg;
  a.dp_slabs = nullptr; a.dp_world = 1; a.mb0 = 0; a.dp_sync = nullptr; a.part_stride = 0; a.dp_uncached = 0;
  {  // OSA_PASS_ONE_XCC=0: the three workgroups on three XCCs (A/B switch; traffic: profiles/r3_pmc_traffic*)
    static const bool one = !(getenv("OSA_PASS_ONE_XCC") && getenv("OSA_PASS_ONE_XCC")[0] == '0');
    a.one_xcc = one ? 1 : 0;
  }
  const int KB = a.nd.KB, OT = a.nd.OUTP / 16;
  hipStream_t st = osa_stream(stream);
  if (ext) {  // extended actor surrogates: single-chunk minibatches only (mask mean / penalty are per minibatch)
    if (B > 64) return OSA_EUNSUPPORTED;
    const bool need_old = ext->kl_coef != 0.f || ext->kl_mask_eta >= 0.f;
    OSA_REQUIRE(!need_old || (ext->old_mean && ext->old_log_std && ext->ld_old_mean >= act_dim));
    if (!need_old && ext->cost_kappa <= 0.f && ext->ratio_scale == 1.f) {
      ext = nullptr;  // nothing extended: the plain instantiation
    } else {
      OSA_REQUIRE(ext->old_mean && ext->old_log_std);  // (the kernel reads them unconditionally)
      a.old_mean = ext->old_mean; a.ld_old_mean = ext->ld_old_mean; a.old_log_std = ext->old_log_std;
      a.ext_kl_coef = ext->kl_coef; a.ext_mask_eta = ext->kl_mask_eta; a.ext_ratio_scale = ext->ratio_scale;
      a.ext_cost_kappa = ext->cost_kappa; a.ext_cost_excess = ext->cost_excess;
#define OSA_PASS_EXT_CASE(K, O) \
  if (KB == K && OT == O) return osa_launch_pass_so<K, O, false, false, true>(a, st)
      OSA_PASS_EXT_CASE(1, 1); OSA_PASS_EXT_CASE(2, 1); OSA_PASS_EXT_CASE(3, 1); OSA_PASS_EXT_CASE(4, 1);
      OSA_PASS_EXT_CASE(5, 1); OSA_PASS_EXT_CASE(6, 1);
      OSA_PASS_EXT_CASE(1, 2); OSA_PASS_EXT_CASE(2, 2); OSA_PASS_EXT_CASE(3, 2); OSA_PASS_EXT_CASE(4, 2);
      OSA_PASS_EXT_CASE(5, 2); OSA_PASS_EXT_CASE(6, 2);
#undef OSA_PASS_EXT_CASE
      return OSA_EUNSUPPORTED;
    }
  }
#define OSA_PASS_CASE(K, O) \
  if (KB == K && OT == O) return (B > 64) ? osa_launch_pass_so<K, O, true>(a, st) : osa_launch_pass_so<K, O, false>(a, st)
  OSA_PASS_CASE(1, 1); OSA_PASS_CASE(2, 1); OSA_PASS_CASE(3, 1); OSA_PASS_CASE(4, 1);
  OSA_PASS_CASE(5, 1); OSA_PASS_CASE(6, 1);
  OSA_PASS_CASE(1, 2); OSA_PASS_CASE(2, 2); OSA_PASS_CASE(3, 2); OSA_PASS_CASE(4, 2);
  OSA_PASS_CASE(5, 2); OSA_PASS_CASE(6, 2);
#undef OSA_PASS_CASE
  return OSA_EUNSUPPORTED;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Data-parallel step without a per-step cross-GPU collective ("replicated data"): every rank holds the
// all-gathered rollout of all W ranks and redundantly computes the whole global optimiser step --
// W workgroups per network produce the W locally clipped gradients (osa_ppo_pass_kernel in dp mode),
// osa_dp_apply_kernel averages them and applies Adam.  Identical arithmetic on every rank => the
// replicas stay bit-identical with no parameter traffic at all; the only collectives left are the
// per-epoch all-gather of the rollout and the scalar statistics.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void osa_dp_apply_kernel(OsaNet nd, float* params, float* adam_m,
                                                           float* adam_v, const int* adam_step,
                                                           const float* slabs, int W, OsaPassHp hp,
                                                           const float* lr_dev, int nets_mask,
                                                           float* stats, int step_index) {
  // grid (ceil(P/256), 3): one parameter per thread, W independent slab loads in flight per thread
  const int net = blockIdx.y;
  if (!((nets_mask >> net) & 1)) return;
  const int P = nd.P, SW = P + PNSTAT;
  const bool critic = net != 0;
  const float* __restrict__ sl = slabs + (long)net * W * SW;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int step = adam_step[net] + step_index + 1;  // adam_step advances once per pass (osa_ppo_dp_end_pass)
  const float lr = lr_dev ? lr_dev[critic ? 1 : 0] : (critic ? hp.lr_critic : hp.lr_actor);
  // 1 - beta^t via exp(t log beta) in float64 (a double pow() per thread costs more than the update)
  const double bc1 = -expm1((double)step * log((double)hp.beta1));
  const double bc2 = -expm1((double)step * log((double)hp.beta2));
  const float step_size = (float)((double)lr / bc1);
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  const float invW = 1.f / (float)W;
  if (e < P) {
    float g = 0.f;
    for (int r = 0; r < W; ++r) g += sl[(long)r * SW + e];  // fixed order: deterministic on every rank
    g *= invW;                                              // avg_grads (distributed.py:193-198)
    if (critic && e >= nd.oLS) g = 0.f;
    const long o = (long)net * P + e;
    float mv = adam_m[o], vv = adam_v[o];
    params[o] = osa_adam_update(g, mv, vv, params[o], hp.beta1, hp.beta2, step_size, inv_bc2_sqrt, hp.adam_eps);
    adam_m[o] = mv;
    adam_v[o] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < 5) {
    const int k = threadIdx.x;
    float t = 0.f;
    for (int r = 0; r < W; ++r) t += sl[(long)r * SW + P + k];
    t *= invW;  // what Logger.get_stats averages across ranks
    // k: 0 loss, 1 ratio, 2 sum p^2, 3 |g|, 4 entropy
    if (net == 0) {
      if (k == 0) stats[2] = t;
      if (k == 1) stats[3] = t;
      if (k == 4) stats[4] = t;
      if (k == 3) stats[7] = t;
    } else {
      if (k == 0) stats[net - 1] = t;
      if (k == 2) stats[4 + net] = t;
      if (k == 3) stats[7 + net] = t;
    }
  }
}

__global__ void osa_dp_step_count_kernel(int* adam_step, int nets_mask, int nsteps) {
  const int net = threadIdx.x;
  if (net < 3 && ((nets_mask >> net) & 1)) adam_step[net] += nsteps;
}

extern "C" {

size_t osa_ppo_dp_ws_floats(int obs_dim, int act_dim, int hidden, int world) {
  if (!osa_ppo_pass_supported(obs_dim, act_dim, hidden) || world < 1) return 0;
  const OsaNet nd = osa_make_net(obs_dim, act_dim, hidden);
  return (size_t)3 * world * (nd.P + PNSTAT);
}

int osa_ppo_dp_step(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                    int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                    const float* logp, const float* target_value_r, const float* target_value_c,
                    const float* adv_r, const float* adv_c, const long* perm, long M, int B, int world,
                    int step_index, const float* lagrange, const osa_ppo_hparams* hp,
                    const float* lr_dev, int loss_kind, int nets_mask, float* slabs,
                    float* step_stats, void* stream) {
  return osa_ppo_dp_step_phase(obs_dim, act_dim, hidden, params, adam_m, adam_v, adam_step, obs, ld_obs, act, ld_act,
                               logp, target_value_r, target_value_c, adv_r, adv_c, perm, M, B, world, step_index,
                               lagrange, hp, lr_dev, loss_kind, nets_mask, slabs, step_stats, 0, stream);
}

int osa_ppo_dp_step_phase(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                          int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                          const float* logp, const float* target_value_r, const float* target_value_c,
                          const float* adv_r, const float* adv_c, const long* perm, long M, int B, int world,
                          int step_index, const float* lagrange, const osa_ppo_hparams* hp,
                          const float* lr_dev, int loss_kind, int nets_mask, float* slabs,
                          float* step_stats, int phase, void* stream) {
  if (!osa_ppo_pass_supported(obs_dim, act_dim, hidden)) return OSA_EUNSUPPORTED;
  OSA_REQUIRE(phase >= 0 && phase <= 2);
  if (phase == 2) {  // Adam from the (all-reduced) slabs only
    OSA_REQUIRE(params && adam_m && adam_v && adam_step && hp && step_stats && slabs && world >= 1 && step_index >= 0);
    const OsaNet nd = osa_make_net(obs_dim, act_dim, hidden);
    OsaPassHp h = {};
    h.lr_actor = hp->lr_actor; h.lr_critic = hp->lr_critic; h.beta1 = hp->beta1; h.beta2 = hp->beta2;
    h.adam_eps = hp->adam_eps;
    hipLaunchKernelGGL(osa_dp_apply_kernel, dim3((nd.P + 255) / 256, 3), dim3(256), 0, osa_stream(stream), nd, params,
                       adam_m, adam_v, adam_step, slabs, world, h, lr_dev, nets_mask & (hp->use_cost ? 7 : 3), step_stats,
                       step_index);
    return hipGetLastError() == hipSuccess ? OSA_OK : OSA_EHIP;
  }
  OSA_REQUIRE(params && adam_m && adam_v && adam_step && obs && act && logp && hp && step_stats && slabs);
  OSA_REQUIRE(target_value_r && target_value_c && adv_r && adv_c && M > 0 && B > 0 && world >= 1);
  OSA_REQUIRE(ld_obs >= obs_dim && ld_act >= act_dim && step_index >= 0 && (long)step_index * B < M);
  if ((double)M * world * ld_obs >= 2147483647.0) return OSA_EUNSUPPORTED;
  if (ld_obs % 4 != 0 || (reinterpret_cast<uintptr_t>(obs) & 15) != 0) return OSA_EUNSUPPORTED;  // pad rows
  OsaPassArgs a = {};
  a.ext_ratio_scale = 1.f; a.ext_mask_eta = -1.f;
  a.nd = osa_make_net(obs_dim, act_dim, hidden);
  a.params = params; a.adam_m = adam_m; a.adam_v = adam_v; a.adam_step = adam_step;
  a.obs = obs; a.ld_obs = ld_obs; a.act = act; a.ld_act = ld_act; a.logp = logp;
  a.tgt_r = target_value_r; a.tgt_c = target_value_c; a.adv_r = adv_r; a.adv_c = adv_c;
  a.perm = perm; a.M = M; a.B = B; a.nmb = 1; a.lagrange = lagrange;
  a.hp.clip = hp->clip; a.hp.entropy_coef = hp->entropy_coef;
  a.hp.critic_norm_coef = hp->critic_norm_coef; a.hp.max_grad_norm = hp->max_grad_norm;
  a.hp.lr_actor = hp->lr_actor; a.hp.lr_critic = hp->lr_critic; a.hp.beta1 = hp->beta1;
  a.hp.beta2 = hp->beta2; a.hp.adam_eps = hp->adam_eps; a.hp.use_critic_norm = hp->use_critic_norm;
  a.hp.use_max_grad_norm = hp->use_max_grad_norm; a.hp.use_cost = hp->use_cost;
  a.loss_kind = loss_kind; a.nets_mask = nets_mask & (hp->use_cost ? 7 : 3); a.stats = step_stats;
  a.dbg = nullptr; a.dp_slabs = slabs; a.dp_world = world; a.mb0 = step_index; a.dp_sync = nullptr; a.part_stride = 0; a.dp_uncached = 0;
  const int KB = a.nd.KB, OT = a.nd.OUTP / 16;
  hipStream_t st = osa_stream(stream);
  int rc = OSA_EUNSUPPORTED;
#define OSA_DP_CASE(K, O)                                                                        \
  if (KB == K && OT == O)                                                                        \
    rc = (B > 64) ? osa_launch_pass<K, O, true, false, false, false, true>(a, st, world)  \
                  : osa_launch_pass<K, O, false, false, false, false, true>(a, st, world)
  OSA_DP_CASE(1, 1); OSA_DP_CASE(2, 1); OSA_DP_CASE(3, 1); OSA_DP_CASE(4, 1); OSA_DP_CASE(5, 1);
  OSA_DP_CASE(6, 1); OSA_DP_CASE(1, 2); OSA_DP_CASE(2, 2); OSA_DP_CASE(3, 2); OSA_DP_CASE(4, 2);
  OSA_DP_CASE(5, 2); OSA_DP_CASE(6, 2);
#undef OSA_DP_CASE
  if (rc != OSA_OK || phase == 1) return rc;  // (phase 1: the gradients only -- the caller's all-reduce comes next)
  hipLaunchKernelGGL(osa_dp_apply_kernel, dim3((a.nd.P + 255) / 256, 3), dim3(256), 0, st, a.nd, params,
                     adam_m, adam_v, adam_step, slabs, world, a.hp, lr_dev, a.nets_mask, step_stats,
                     step_index);
  return hipGetLastError() == hipSuccess ? OSA_OK : OSA_EHIP;
}

// Uncached exchange buffers handed out by osa_dp_exchange_alloc (base, bytes)
static const int OSA_MAX_XCH = 1024;  // (uncached buffers alive at once: one or two per updater; Python frees them when the updater is collected)
static char* g_xch_base[OSA_MAX_XCH];
static size_t g_xch_bytes[OSA_MAX_XCH];

// (library-internal: also used by wide_split_kernel.hip; not declared in the public header)
bool osa_is_exchange_ptr(const void* p) {
  const char* c = static_cast<const char*>(p);
  for (int k = 0; k < OSA_MAX_XCH; ++k)
    if (g_xch_base[k] && c >= g_xch_base[k] && c < g_xch_base[k] + g_xch_bytes[k]) return true;
  return false;
}

int osa_dp_exchange_alloc(size_t floats, float** out) {
  OSA_REQUIRE(out != nullptr && floats > 0);
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, floats * sizeof(float), hipDeviceMallocUncached) != hipSuccess || !p) {
    (void)hipGetLastError();
    return OSA_EUNSUPPORTED;
  }
  if (hipMemset(p, 0, floats * sizeof(float)) != hipSuccess) return OSA_EHIP;
  for (int k = 0; k < OSA_MAX_XCH; ++k)
    if (!g_xch_base[k]) {
      g_xch_base[k] = static_cast<char*>(p);
      g_xch_bytes[k] = floats * sizeof(float);
      *out = static_cast<float*>(p);
      return OSA_OK;
    }
  (void)hipFree(p);
  return OSA_EUNSUPPORTED;
}

int osa_dp_exchange_free(float* p) {
  for (int k = 0; k < OSA_MAX_XCH; ++k)
    if (g_xch_base[k] == reinterpret_cast<char*>(p)) {
      g_xch_base[k] = nullptr;
      return hipFree(p) == hipSuccess ? OSA_OK : OSA_EHIP;
    }
  return OSA_EINVAL;
}

static size_t osa_dp_pass_xs(const OsaNet& nd) {
  return (size_t)(4 + nd.KB + nd.OUTP / 16) * 1024 + 256 + PNSTAT;
}

size_t osa_ppo_dp_pass_ws_floats(int obs_dim, int act_dim, int hidden, int world) {
  if (!osa_ppo_pass_supported(obs_dim, act_dim, hidden) || world < 1) return 0;
  const OsaNet nd = osa_make_net(obs_dim, act_dim, hidden);
  return (size_t)2 * 3 * world * osa_dp_pass_xs(nd) + (size_t)2 * 3 * osa_dp_pass_xs(nd);  // (+ one spare slab set)
}

int osa_ppo_dp_pass(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                    int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                    const float* logp, const float* target_value_r, const float* target_value_c,
                    const float* adv_r, const float* adv_c, const long* perm, long M, int B, int world,
                    const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                    float* exchange, int* sync, float* step_stats, void* stream) {
  return osa_ppo_dp_pass_placed(obs_dim, act_dim, hidden, params, adam_m, adam_v, adam_step, obs, ld_obs, act,
                                ld_act, logp, target_value_r, target_value_c, adv_r, adv_c, perm, M, B, world,
                                lagrange, hp, loss_kind, nets_mask, exchange, sync, 0, step_stats, stream);
}

static int osa_coop_pass(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                         int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                         const float* logp, const float* target_value_r, const float* target_value_c,
                         const float* adv_r, const float* adv_c, const long* perm, long M, int B, int world,
                         const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                         float* exchange, int* sync, int local, int chunk, int ranks, float* step_stats, void* stream);

int osa_ppo_dp_pass_placed(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                           int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                           const float* logp, const float* target_value_r, const float* target_value_c,
                           const float* adv_r, const float* adv_c, const long* perm, long M, int B, int world,
                           const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                           float* exchange, int* sync, int local, float* step_stats, void* stream) {
  return osa_coop_pass(obs_dim, act_dim, hidden, params, adam_m, adam_v, adam_step, obs, ld_obs, act, ld_act, logp,
                       target_value_r, target_value_c, adv_r, adv_c, perm, M, B, world, lagrange, hp, loss_kind,
                       nets_mask, exchange, sync, local, 0, 1, step_stats, stream);
}

int osa_ppo_chunked_pass(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                         int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                         const float* logp, const float* target_value_r, const float* target_value_c,
                         const float* adv_r, const float* adv_c, const long* perm, long M, int B,
                         const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                         float* exchange, int* sync, int local, float* step_stats, void* stream) {
  if (B <= 64 || B > 64 * 32) return OSA_EUNSUPPORTED;  // one chunk: osa_ppo_pass
  return osa_coop_pass(obs_dim, act_dim, hidden, params, adam_m, adam_v, adam_step, obs, ld_obs, act, ld_act, logp,
                       target_value_r, target_value_c, adv_r, adv_c, perm, M, B, (B + 63) / 64, lagrange, hp,
                       loss_kind, nets_mask, exchange, sync, local, 1, 1, step_stats, stream);
}

size_t osa_ppo_dp_chunked_pass_ws_floats(int obs_dim, int act_dim, int hidden, int B, int world) {
  if (!osa_ppo_pass_supported(obs_dim, act_dim, hidden) || world < 1 || B <= 64) return 0;
  const OsaNet nd = osa_make_net(obs_dim, act_dim, hidden);
  const size_t chunks = (size_t)(B + 63) / 64;
  return (size_t)2 * 3 * (world * chunks) * osa_dp_pass_xs(nd) + (size_t)2 * 3 * world * osa_dp_pass_xs(nd);
}

int osa_ppo_dp_chunked_pass(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                            int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                            const float* logp, const float* target_value_r, const float* target_value_c,
                            const float* adv_r, const float* adv_c, const long* perm, long M, int B, int world,
                            const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                            float* exchange, int* sync, int local, float* step_stats, void* stream) {
  if (B <= 64 || B > 64 * 32 || world < 1 || world > 16) return OSA_EUNSUPPORTED;
  const int chunks = (B + 63) / 64;
  return osa_coop_pass(obs_dim, act_dim, hidden, params, adam_m, adam_v, adam_step, obs, ld_obs, act, ld_act, logp,
                       target_value_r, target_value_c, adv_r, adv_c, perm, M, B, world * chunks, lagrange, hp,
                       loss_kind, nets_mask, exchange, sync, local, 1, world, step_stats, stream);
}

static int osa_coop_pass(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                         int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                         const float* logp, const float* target_value_r, const float* target_value_c,
                         const float* adv_r, const float* adv_c, const long* perm, long M, int B, int world,
                         const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                         float* exchange, int* sync, int local, int chunk, int ranks, float* step_stats, void* stream) {
  if (!osa_ppo_pass_supported(obs_dim, act_dim, hidden)) return OSA_EUNSUPPORTED;
  if (local && osa_is_exchange_ptr(exchange)) return OSA_EINVAL;  // the XCC's L2 serves ordinary memory
  OSA_REQUIRE(params && adam_m && adam_v && adam_step && obs && act && logp && hp && step_stats);
  OSA_REQUIRE(target_value_r && target_value_c && adv_r && adv_c && M > 0 && B > 0 && world >= 1);
  OSA_REQUIRE(exchange && sync && ld_obs >= obs_dim && ld_act >= act_dim);
  if ((double)M * (chunk ? ranks : world) * ld_obs >= 2147483647.0) return OSA_EUNSUPPORTED;
  if (ld_obs % 4 != 0 || (reinterpret_cast<uintptr_t>(obs) & 15) != 0) return OSA_EUNSUPPORTED;  // pad rows
  // all 3 * world workgroups must be co-resident (one per compute unit: ~150 KB of LDS each)
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return OSA_EHIP;
  if (3 * world > cus || (local && world > cus / 8)) return OSA_EUNSUPPORTED;
  OsaPassArgs a = {};
  a.dp_local = local;  // 0, 1, or 3 (see the kernel)
  a.dp_chunk = chunk ? 1 : 0;
  a.dp_ranks = chunk ? ranks : 1;
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
  a.dbg = g_osa_pass_dbg; a.dp_slabs = exchange; a.dp_world = world; a.mb0 = 0; a.dp_sync = sync; a.part_stride = 0; a.dp_uncached = osa_is_exchange_ptr(exchange) ? 1 : 0;
  hipStream_t st = osa_stream(stream);
  // sync[0..2]: per-network arrival counters of this pass; sync[3]: STICKY "a peer never arrived" flag --
  // it survives the per-pass reset so that a timeout in any pass of an update is still visible when the
  // host reads it (the caller zeroes all four words once, at allocation)
  if (hipMemsetAsync(sync, 0, 3 * sizeof(int), st) != hipSuccess) return OSA_EHIP;
  if (local && hipMemsetAsync(sync + 4, 0, 4 * sizeof(int), st) != hipSuccess) return OSA_EHIP;  // XCC masks, arrivals
  // (chunk mode under data parallelism: the per-rank arrival counters of the first hand-off; sync is int[64] there)
  if ((chunk && ranks > 1) && hipMemsetAsync(sync + 8, 0, 48 * sizeof(int), st) != hipSuccess) return OSA_EHIP;
  const int KB = a.nd.KB, OT = a.nd.OUTP / 16;
#define OSA_DPP_CASE(K, O)                                                                       \
  if (KB == K && OT == O)                                                                        \
    return (chunk && ranks > 1) ? osa_launch_pass_so<K, O, false, true, false, true>(a, st, world) \
           : (B > 64 && !chunk) ? osa_launch_pass<K, O, true, true>(a, st, world)                \
                                : osa_launch_pass_so<K, O, false, true>(a, st, world)
  OSA_DPP_CASE(1, 1); OSA_DPP_CASE(2, 1); OSA_DPP_CASE(3, 1); OSA_DPP_CASE(4, 1); OSA_DPP_CASE(5, 1);
  OSA_DPP_CASE(6, 1); OSA_DPP_CASE(1, 2); OSA_DPP_CASE(2, 2); OSA_DPP_CASE(3, 2); OSA_DPP_CASE(4, 2);
  OSA_DPP_CASE(5, 2); OSA_DPP_CASE(6, 2);
#undef OSA_DPP_CASE
  return OSA_EUNSUPPORTED;
}

}  // extern "C"

// Large-batch gradient on the persistent kernel's machinery (called by osa_ppo_minibatch_ext): nblk
// workgroups per network, LDS-resident weights, register accumulators, one raw partial-gradient slab each.
int osa_pass_partial_grad(int obs_dim, int act_dim, int hidden, float* params, const float* obs, int ld_obs,
                          const float* act, int ld_act, const float* logp, const float* target_value_r,
                          const float* target_value_c, const float* adv_r, const float* adv_c,
                          const long* idx, int B, const float* lagrange, const osa_ppo_hparams* hp,
                          int loss_kind, int nets_mask, int nblk, float* slabs, void* stream) {
  if (!osa_ppo_pass_supported(obs_dim, act_dim, hidden) || B <= 64) return OSA_EUNSUPPORTED;
  if (ld_obs % 4 != 0 || (reinterpret_cast<uintptr_t>(obs) & 15) != 0) return OSA_EUNSUPPORTED;
  OsaPassArgs a = {};
  a.ext_ratio_scale = 1.f; a.ext_mask_eta = -1.f;
  a.nd = osa_make_net(obs_dim, act_dim, hidden);
  a.params = params; a.adam_m = params; a.adam_v = params; a.adam_step = nullptr;  // untouched in this mode
  a.obs = obs; a.ld_obs = ld_obs; a.act = act; a.ld_act = ld_act; a.logp = logp;
  a.tgt_r = target_value_r; a.tgt_c = target_value_c; a.adv_r = adv_r; a.adv_c = adv_c;
  a.perm = idx; a.M = B; a.B = B; a.nmb = 1; a.lagrange = lagrange;
  a.hp.clip = hp->clip; a.hp.entropy_coef = hp->entropy_coef;
  a.hp.critic_norm_coef = hp->critic_norm_coef; a.hp.max_grad_norm = hp->max_grad_norm;
  a.hp.lr_actor = hp->lr_actor; a.hp.lr_critic = hp->lr_critic; a.hp.beta1 = hp->beta1;
  a.hp.beta2 = hp->beta2; a.hp.adam_eps = hp->adam_eps; a.hp.use_critic_norm = hp->use_critic_norm;
  a.hp.use_max_grad_norm = hp->use_max_grad_norm; a.hp.use_cost = hp->use_cost;
  a.loss_kind = loss_kind; a.nets_mask = nets_mask; a.stats = nullptr;
  a.dbg = nullptr; a.dp_slabs = slabs; a.dp_world = nblk; a.mb0 = 0; a.dp_sync = nullptr;
  a.part_stride = nblk; a.dp_uncached = 0;
  const int KB = a.nd.KB, OT = a.nd.OUTP / 16;
  hipStream_t st = osa_stream(stream);
#define OSA_PG_CASE(K, O) \
  if (KB == K && OT == O) return osa_launch_pass<K, O, true, false, false, false, true>(a, st, nblk)
  OSA_PG_CASE(1, 1); OSA_PG_CASE(2, 1); OSA_PG_CASE(3, 1); OSA_PG_CASE(4, 1); OSA_PG_CASE(5, 1);
  OSA_PG_CASE(6, 1); OSA_PG_CASE(1, 2); OSA_PG_CASE(2, 2); OSA_PG_CASE(3, 2); OSA_PG_CASE(4, 2);
  OSA_PG_CASE(5, 2); OSA_PG_CASE(6, 2);
#undef OSA_PG_CASE
  return OSA_EUNSUPPORTED;
}

extern "C" {

int osa_ppo_dp_end_pass(int* adam_step, int nets_mask, int nsteps, void* stream) {
  OSA_REQUIRE(adam_step && nsteps > 0);
  hipLaunchKernelGGL(osa_dp_step_count_kernel, dim3(1), dim3(64), 0, osa_stream(stream), adam_step,
                     nets_mask, nsteps);
  OSA_CHECK_LAUNCH();
  return OSA_OK;
}

}  // extern "C"
