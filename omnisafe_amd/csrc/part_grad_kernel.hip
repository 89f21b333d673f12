// Balanced partial gradients of one LARGE minibatch (B >= 2048) on the persistent pass kernel's machinery (round 4).
// Its own translation unit: the 18 instantiations compile in parallel with ppo_pass_kernel.hip's ~90.
#define OSA_BODY_OPAQUE_TID 1
#define OSA_BODY_PART_ONLY 1
#ifdef OSA_PART_CLOCKS  // tools/build_variant_lib.sh pclocks part_grad_kernel.hip -DOSA_PART_CLOCKS (tools/part_kernel_timeline.py)
#define OSA_PART_MARK(k)                                                                              \
  do {                                                                                                \
    if (a.dbg && threadIdx.x == 0) a.dbg[16 * blockIdx.x + (k)] = (long long)__builtin_amdgcn_s_memtime(); \
  } while (0)
#endif
#include "ppo_pass_body.h"

static size_t osa_part_lds_bytes(int KB, int OT) {
  const size_t fl = (size_t)osa_pass_lds_floats(KB, OT) + (osa_pass_has_w2t(KB, OT) ? 64 * PSLD : 0);
  return fl * sizeof(float);
}

// Balanced partial gradients of ONE large minibatch (OsaPassArgs.part_tpw): see there.
template <int KB, int OT, bool SO>
__global__ __launch_bounds__(256, 1) void osa_ppo_part_kernel(OsaPassArgs a) {
  const int nchunk = (a.B + 63) / 64;
  const int t_lo = blockIdx.x * a.part_tpw, t_hi = t_lo + a.part_tpw;  // this workgroup's tasks [t_lo, t_hi)
  // debug clocks (osa_debug_set_part_clock_buffer): per workgroup {start, end} on the 100 MHz constant clock (comparable
  // across compute units: dispatch skew, per-workgroup duration) and {start, end} shader cycles
  if (a.dbg && threadIdx.x == 0) {
    a.dbg[16 * blockIdx.x + 0] = (long long)__builtin_amdgcn_s_memrealtime();
    a.dbg[16 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memtime();
  }
#if defined(OSA_PART_V0)
  {
    const int net = t_lo / nchunk, n_lo = net * nchunk;
    const int hi = t_hi < n_lo + nchunk ? t_hi : n_lo + nchunk;
    osa_ppo_pass_body<KB, OT, true, false, false, false, true, SO>(a, net, blockIdx.x - n_lo / a.part_tpw,
                                                                                     t_lo - n_lo, hi - t_lo);
    return;
  }
#endif
  int q = 0;  // position of the network among the launch's networks
#pragma nounroll
  for (int net = 0; net < 3; ++net) {
    if (!((a.nets_mask >> net) & 1)) continue;
    const int n_lo = q * nchunk, n_hi = n_lo + nchunk;  // the network's tasks
    const int lo = t_lo > n_lo ? t_lo : n_lo, hi = t_hi < n_hi ? t_hi : n_hi;
    if (lo < hi) {  // (block-uniform)
      const int wfirst = n_lo / a.part_tpw;  // first workgroup with a task of this network: slab 0
      osa_ppo_pass_body<KB, OT, true, false, false, false, true, SO>(a, net, blockIdx.x - wfirst,
                                                                                       lo - n_lo, hi - lo);
      __syncthreads();  // the next segment reloads the LDS master copy
    }
    ++q;
  }
  if (a.dbg && threadIdx.x == 0) {
    a.dbg[16 * blockIdx.x + 2] = (long long)__builtin_amdgcn_s_memrealtime();
    a.dbg[16 * blockIdx.x + 3] = (long long)__builtin_amdgcn_s_memtime();
  }
}


template <int KB, int OT, bool SO>
static int osa_launch_part(const OsaPassArgs& a, hipStream_t stream, int G) {
  static OsaPerDeviceOnce attr_set;
  const size_t lds = osa_part_lds_bytes(KB, OT);
  if (lds > 160 * 1024) return OSA_EUNSUPPORTED;
  if (attr_set.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&osa_ppo_part_kernel<KB, OT, SO>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return OSA_EHIP;
    attr_set.set();
  }
  hipLaunchKernelGGL((osa_ppo_part_kernel<KB, OT, SO>), dim3(G), dim3(256), lds, stream, a);
  return hipGetLastError() == hipSuccess ? OSA_OK : OSA_EHIP;
}

static long long* g_osa_part_dbg = nullptr;
extern "C" int osa_debug_set_part_clock_buffer(long long* dev_ptr) {
  g_osa_part_dbg = dev_ptr;
  return OSA_OK;
}

// Balanced form of osa_pass_partial_grad: G <= max_wg workgroups share the chunk-tasks of all networks of the mask
// evenly (contiguous ranges, OsaPassArgs.part_tpw).  Outputs: nslab[net] = slabs written for network net (0 for
// networks outside the mask), *stride = slab stride per network in `slabs` ([3][stride][P + 16]).
int osa_pass_partial_grad_balanced(int obs_dim, int act_dim, int hidden, float* params, const float* obs, int ld_obs,
                                   const float* act, int ld_act, const float* logp, const float* target_value_r,
                                   const float* target_value_c, const float* adv_r, const float* adv_c,
                                   const long* idx, int B, const float* lagrange, const osa_ppo_hparams* hp,
                                   int loss_kind, int nets_mask, int max_wg, int max_stride, float* slabs,
                                   int* nslab, int* stride, void* stream) {
  if (!osa_ppo_pass_supported(obs_dim, act_dim, hidden) || B <= 64) return OSA_EUNSUPPORTED;
  if (ld_obs % 4 != 0 || (reinterpret_cast<uintptr_t>(obs) & 15) != 0) return OSA_EUNSUPPORTED;
  const int nchunk = (B + 63) / 64, nq = __builtin_popcount(nets_mask & 7);
  if (nq == 0 || max_wg < 1) return OSA_EUNSUPPORTED;
  const long ntasks = (long)nq * nchunk;
  int G = (int)(ntasks < max_wg ? ntasks : max_wg);
  const int tpw = (int)((ntasks + G - 1) / G);
  G = (int)((ntasks + tpw - 1) / tpw);
  int q = 0, st = 0;
  for (int net = 0; net < 3; ++net) {
    nslab[net] = 0;
    if (!((nets_mask >> net) & 1)) continue;
    const int wfirst = (q * nchunk) / tpw, wlast = ((q + 1) * nchunk - 1) / tpw;
    nslab[net] = wlast - wfirst + 1;
    if (nslab[net] > st) st = nslab[net];
    ++q;
  }
  if (st > max_stride) return OSA_EUNSUPPORTED;
  *stride = st;
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
  a.dbg = g_osa_part_dbg; a.dp_slabs = slabs; a.dp_world = st; a.mb0 = 0; a.dp_sync = nullptr;
  a.part_stride = -1; a.part_tpw = tpw; a.dp_uncached = 0;
  const int KB = a.nd.KB, OT = a.nd.OUTP / 16;
  hipStream_t s = osa_stream(stream);
#define OSA_PB_CASE(K, O)                                                                  \
  if (KB == K && OT == O) {                                                                \
    if constexpr (O == 1) {                                                                \
      if (a.nd.act_dim <= 2) return osa_launch_part<K, O, true>(a, s, G);                  \
    }                                                                                      \
    return osa_launch_part<K, O, false>(a, s, G);                                          \
  }
#ifdef OSA_PART_QUICK  // (one instantiation: register-pressure experiments)
  OSA_PB_CASE(4, 1)
#else
  OSA_PB_CASE(1, 1) OSA_PB_CASE(2, 1) OSA_PB_CASE(3, 1) OSA_PB_CASE(4, 1) OSA_PB_CASE(5, 1) OSA_PB_CASE(6, 1)
  OSA_PB_CASE(1, 2) OSA_PB_CASE(2, 2) OSA_PB_CASE(3, 2) OSA_PB_CASE(4, 2) OSA_PB_CASE(5, 2) OSA_PB_CASE(6, 2)
#endif
#undef OSA_PB_CASE
  return OSA_EUNSUPPORTED;
}

