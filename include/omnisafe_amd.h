/* omnisafe_amd.h -- C ABI of libomnisafe_amd.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE hot path of PKU-Alignment/omnisafe: on-policy rollout ->
 * VectorOnPolicyBuffer (dual reward+cost GAE) -> PPOLag / TRPOLag / CPO update.
 * The reference has no FFI for this path (it is pure Python on torch); each
 * entry point below therefore cites the reference Python function whose arithmetic it replaces
 * (paths relative to the reference repository root).  The Python host side (the omnisafe_amd package) binds
 * these symbols with ctypes and mirrors the reference's classes; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name ends in _host; the caller owns all memory;
 *    nothing is allocated or freed inside the library;
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream); all calls are
 *    asynchronous w.r.t. the host and never synchronise;
 *  - return value: 0 = OSA_OK, negative = error (osa_strerror); no exceptions cross the boundary;
 *  - all floating point data is IEEE float32 unless declared double; flags are uint8;
 *  - the (T, N) rollout buffer is TIME-MAJOR: element (t, n) of a per-step scalar lives at [t*N + n],
 *    rows of obs/act at [(t*N + n) * ld].  VectorOnPolicyBuffer.get()'s ENV-MAJOR order
 *    (sample i = n*T + t, omnisafe/common/buffer/vector_onpolicy_buffer.py:125-129) is produced by
 *    osa_buffer_get.
 *  - one host thread per device context (the reference is single-threaded per rank).
 */
#ifndef OMNISAFE_AMD_H
#define OMNISAFE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSA_OK 0
#define OSA_EINVAL (-1)       /* bad argument (null pointer, non-positive size, unsupported dims) */
#define OSA_EHIP (-2)         /* a HIP runtime call / kernel launch failed */
#define OSA_EUNSUPPORTED (-3) /* valid in the reference, not implemented here (yet) */

/* advantage estimators: omnisafe/common/buffer/onpolicy_buffer.py:299-331 */
#define OSA_EST_GAE 0
#define OSA_EST_GAE_RTG 1
#define OSA_EST_PLAIN 2
#define OSA_EST_VTRACE 3 /* onpolicy_buffer.py:312-326 -> _calculate_v_trace :338-405; the reference passes
                            the same probabilities as policy and behaviour, i.e. rho = c = 1 */

const char* osa_strerror(int code);
int osa_version(void);            /* ABI version, currently 1 */
const char* osa_build_arch(void); /* "gfx950" */
/* sha256 (first 32 hex digits) of this header and of the kernel sources the library was built from; the
 * Python binding refuses a library whose digest differs from the sources next to it. */
const char* osa_abi_digest(void);

/* ------------------------------------------------------------------------------------------------
 * Rollout buffer (replaces omnisafe/common/buffer/{onpolicy_buffer,vector_onpolicy_buffer}.py)
 * ---------------------------------------------------------------------------------------------- */

/* VectorOnPolicyBuffer.store (vector_onpolicy_buffer.py:96-99 -> onpolicy_buffer.py:143-146):
 * copies the 7 per-step fields of all N envs into row t of the time-major buffer.  obs/act rows
 * have obs_dim/act_dim valid floats and leading dimensions ld_src (source) / ld_buf (buffer). */
int osa_buffer_store_step(int t, int N, int obs_dim, int act_dim,
                          const float* obs, int ld_obs_src, const float* act, int ld_act_src,
                          const float* reward, const float* cost, const float* value_r,
                          const float* value_c, const float* logp,
                          float* buf_obs, int ld_obs_buf, float* buf_act, int ld_act_buf,
                          float* buf_reward, float* buf_cost, float* buf_value_r, float* buf_value_c,
                          float* buf_logp, void* stream);

/* OnPolicyBuffer.finish_path for every path of every env in ONE backward sweep
 * (onpolicy_buffer.py:170-203, _calculate_adv_and_value_targets :299-331, discount_cumsum
 * omnisafe/utils/math.py:76-82).  path_end[t*N+n] != 0 marks the last step of a path whose bootstrap
 * values boot_r/boot_c[t*N+n] are the arguments finish_path(last_value_r, last_value_c, idx=n)
 * received.  Arithmetic is the reference's exactly: delta in float32 (float32-rounded gamma,
 * separately rounded mul/add/sub), recurrences in float64 with unfused multiply-add, results rounded
 * to float32 on store -> bit-exact.  Outputs are time-major (T, N). */
int osa_gae_scan(const float* reward, const float* cost, const float* value_r, const float* value_c,
                 const uint8_t* path_end, const float* boot_r, const float* boot_c, int T, int N,
                 double gamma, double lam, double lam_c, float penalty_coef, int estimator,
                 float* adv_r, float* adv_c, float* target_value_r, float* target_value_c,
                 float* discounted_ret, void* stream);

/* The same sweep parallel over TIME -- the wavefront backward scan with LDS staging: a workgroup owns 16
 * consecutive envs, walks the time axis backwards in tiles of 64 steps staged in LDS with coalesced row
 * segments, and scans each env's 64 steps across the 64 lanes of a wave (Hillis-Steele scan of the affine
 * maps y_t = x_t + c_t y_{t+1}, carries chained from tile to tile).  For few envs / long horizons (e.g.
 * vector_env_nums = 4, T = 5000) where osa_gae_scan leaves the chip idle; the caller picks by (T, N).
 * Same float32 deltas; the float64 recurrences are associated as a tree instead of a chain, so outputs equal
 * osa_gae_scan's to float64 rounding before the final float32 rounding (tests: rtol 1e-5).  Estimators gae,
 * gae-rtg, plain (v-trace is a float32 chain: OSA_EUNSUPPORTED, use osa_gae_scan).
 * Replaces: the same reference code as osa_gae_scan. */
int osa_gae_scan_tiled(const float* reward, const float* cost, const float* value_r, const float* value_c,
                       const uint8_t* path_end, const float* boot_r, const float* boot_c, int T, int N,
                       double gamma, double lam, double lam_c, float penalty_coef, int estimator,
                       float* adv_r, float* adv_c, float* target_value_r, float* target_value_c,
                       float* discounted_ret, void* stream);
/* The same scan SPLIT OVER TIME across workgroups (csrc/buffer_kernels.hip K5c, round 3): levels of 128 steps, a
 * wave = 64 envs x 16 steps with its inputs in registers, per-env carries chained between levels by a decoupled
 * look-back (8-byte agent-scope words in `ws`; a level that contains a path end publishes its carry at once, so
 * chains end at episode boundaries).  (T / 16) x (N / 64) waves whatever the shape: the large-N / long-T buffers
 * where osa_gae_scan has too few lanes in flight and osa_gae_scan_tiled pays its log-depth scan.  Arithmetic: the
 * sequential kernel's step for step given the incoming carry; the carry is assembled by the affine identity
 * (float64 re-association): same tolerance class as osa_gae_scan_tiled, deterministic (independent of timing).
 * ws: osa_gae_chained_ws_doubles(T, N) doubles; the carries and the ticket are initialised by every call, the LAST
 * double is a sticky time-out word the caller zeroes ONCE after allocation: a lane that gives up waiting for a
 * carry (bounded spin: a device shared with / preempted by another long-running kernel) sets it and its outputs
 * are NaN -- osa_gae_chained_timed_out(ws, T, N, &flag) reads it (synchronous 4-byte copy; 0 = fine), which the
 * caller does at its next host synchronisation.  v-trace: OSA_EUNSUPPORTED. */
size_t osa_gae_chained_ws_doubles(int T, int N);
int osa_gae_chained_timed_out(const double* ws, int T, int N, int* out);
int osa_gae_scan_chained(const float* reward, const float* cost, const float* value_r, const float* value_c,
                         const uint8_t* path_end, const float* boot_r, const float* boot_c, int T, int N,
                         double gamma, double lam, double lam_c, float penalty_coef, int estimator,
                         float* adv_r, float* adv_c, float* target_value_r, float* target_value_c,
                         float* discounted_ret, double* ws, void* stream);

/* Advantage statistics of VectorOnPolicyBuffer.get (vector_onpolicy_buffer.py:131-136 ->
 * omnisafe/utils/distributed.py:382-392), split in two phases so the cross-rank all-reduce (RCCL) can
 * sit between them.  stats is 8 doubles on the device:
 *   [0] sum(adv_r) [1] sum(adv_c) [2] n   -- written by phase 1 (local); all-reduce(SUM) [0..2]
 *   [3] sum((adv_r - mean_r)^2)            -- written by phase 2 (local); all-reduce(SUM) [3]
 *   [4] mean_r [5] mean_c (float32-rounded, as the reference's float32 tensors) -- phase 2
 *   [6] std_r = sqrt([3]/[2]) rounded to float32 -- written by osa_buffer_get
 * ws: workspace of at least osa_reduce_ws_bytes() bytes. */
size_t osa_reduce_ws_bytes(void);
int osa_adv_stats_phase1(const float* adv_r, const float* adv_c, long M, double* ws, double* stats,
                         void* stream);
int osa_adv_stats_phase2(const float* adv_r, long M, double* ws, double* stats, void* stream);

/* VectorOnPolicyBuffer.get (vector_onpolicy_buffer.py:113-138): time-major (T,N) -> env-major (N*T)
 * copy of the 8 returned tensors with adv_r <- (adv_r - mean)/(std + 1e-8) if standardize_r and
 * adv_c <- adv_c - mean_c if standardize_c.  Any source/destination pair may be NULL (skipped). */
int osa_buffer_get(int T, int N, int obs_dim, int act_dim,
                   const float* obs, int ld_obs, const float* act, int ld_act, const float* logp,
                   const float* target_value_r, const float* target_value_c, const float* adv_r,
                   const float* adv_c, const float* discounted_ret, double* stats,
                   int standardize_r, int standardize_c,
                   float* out_obs, int ld_out_obs, float* out_act, int ld_out_act, float* out_logp,
                   float* out_target_value_r, float* out_target_value_c, float* out_adv_r,
                   float* out_adv_c, float* out_discounted_ret, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Actor-critic networks (replaces the torch modules + optimisers of
 * omnisafe/models/actor_critic/{actor_critic,constraint_actor_critic}.py on this path)
 *
 * Three MLPs obs_dim -> H -> H -> {act_dim | 1 | 1} (omnisafe/utils/model.py:73-111).  The `hidden` argument of
 * every entry point carries the width H in its low 16 bits (hidden_sizes [H, H]: 64 = every on-policy YAML
 * default; 32, 128 and 256 on the per-step entry points; anything else: OSA_EUNSUPPORTED) and the hidden
 * activation in bits 16-19 (model_cfgs.*.activation,
 * utils/model.py:47-70): 0 tanh (default: `hidden = 64`), 1 relu, 2 sigmoid, 3 softplus, 4 identity.  The per-step
 * entry points (policy step, minibatch, KL, evaluation, Fisher-vector product) implement all five; the persistent
 * passes (osa_ppo_pass / _wide_ / _split_ / _dp_ / _chunked_) are 64-wide tanh only and report 0 from their *_supported
 * probes for anything else, which routes the update to the per-step entry points.  Each network's parameters, Adam moments and
 * gradients live in one PADDED float32 block of P floats described by osa_mlp_layout:
 *   W1 [H][INP] | b1 [H] | W2 [H][H] | b2 [H] | W3 [OUTP][H] | b3 [OUTP] | log_std [OUTP]
 * (INP/OUTP = obs_dim / max(act_dim,1) rounded up to 16; padding is zero and stays zero).  The three
 * blocks are stored back to back: params[3][P] = actor, reward critic, cost critic.  Conversion to
 * the reference's state_dict tensors / flat-parameter order (omnisafe/utils/tools.py:35-129) is an
 * index map built from osa_mlp_layout on the host.
 * ---------------------------------------------------------------------------------------------- */

/* out12 = {INP, OUTP, oW1, ob1, oW2, ob2, oW3, ob3, oLS, P, H, KB} (offsets in floats). */
int osa_mlp_layout(int obs_dim, int act_dim, int hidden, int* out12);

/* ConstraintActorCritic.step (constraint_actor_critic.py:102-109; GaussianLearningActor.predict /
 * log_prob gaussian_learning_actor.py:81-129; VCritic.forward v_critic.py:89-92) for N observation
 * rows: act = mean + exp(log_std) * eps (or mean if deterministic), logp = sum_d Normal.log_prob,
 * value_r, value_c.  eps: optional external standard-normal noise [N][act_dim] (parity tests);
 * when NULL the kernel draws Philox4x32-10 + Box-Muller normals keyed by (seed, offset + *offset_base, row,
 * dim).  offset_base (device pointer, may be NULL = 0) is the part of the stream position that lives in
 * device memory: a captured hipGraph of a whole rollout replays with frozen by-value offsets 1..T and the
 * caller advances *offset_base by T between replays.
 * nets_mask: bit0 actor, bit1 reward critic, bit2 cost critic (bootstrap calls need critics only).
 * Any output pointer may be NULL.  mean_out optionally receives the distribution mean. */
int osa_policy_step(int obs_dim, int act_dim, int hidden, const float* params, const float* obs,
                    int ld_obs, int N, const float* eps, unsigned long long seed,
                    unsigned long long offset, const unsigned long long* offset_base, int deterministic,
                    int nets_mask, float* act, int ld_act, float* value_r, float* value_c, float* logp,
                    float* mean_out, int ld_mean, void* stream);

/* osa_policy_step with ActionScale.step (omnisafe/envs/wrapper.py:510-514) applied to the sampled action in the same
 * launch: act_env[n][d] = old_min[d] + (old_max[d] - old_min[d]) * (act[n][d] - min_action) / (max_action - min_action)
 * -- the bits of osa_action_scale on the same action.  act_env may be NULL (then exactly osa_policy_step). */
int osa_policy_step_scaled(int obs_dim, int act_dim, int hidden, const float* params, const float* obs,
                           int ld_obs, int N, const float* eps, unsigned long long seed,
                           unsigned long long offset, const unsigned long long* offset_base, int deterministic,
                           int nets_mask, float* act, int ld_act, float* value_r, float* value_c, float* logp,
                           float* mean_out, int ld_mean, float* act_env, int ld_env, const float* old_min,
                           const float* old_max, float min_action, float max_action, void* stream);

/* Hyper-parameters of one optimiser step; field names follow algo_cfgs / model_cfgs of
 * omnisafe/configs/on-policy/PPOLag.yaml. */
typedef struct osa_ppo_hparams {
  float clip;             /* algo_cfgs.clip */
  float entropy_coef;     /* algo_cfgs.entropy_coef */
  float critic_norm_coef; /* algo_cfgs.critic_norm_coef */
  float max_grad_norm;    /* algo_cfgs.max_grad_norm */
  float lr_actor;         /* model_cfgs.actor.lr x LinearLR factor of the current epoch */
  float lr_critic;        /* model_cfgs.critic.lr */
  float beta1, beta2, adam_eps; /* torch.optim.Adam defaults 0.9, 0.999, 1e-8 */
  int use_critic_norm;    /* algo_cfgs.use_critic_norm */
  int use_max_grad_norm;  /* algo_cfgs.use_max_grad_norm */
  int use_cost;           /* algo_cfgs.use_cost */
  const float* lr_device; /* optional: { lr_actor, lr_critic } in DEVICE memory, read at execution time instead of the
                           * two by-value fields -- lets a captured hipGraph of optimiser steps be replayed across
                           * epochs while the LinearLR schedule moves.  Honoured by osa_ppo_minibatch(_ext) and
                           * osa_adam_apply; NULL everywhere else */
} osa_ppo_hparams;

/* One minibatch step of PolicyGradient._update's inner loop for all three networks in one launch:
 * _update_reward_critic / _update_cost_critic (omnisafe/algorithms/on_policy/base/policy_gradient.py:
 * 428-445, 468-485: MSE + critic_norm_coef * sum p^2, clip_grad_norm_, Adam) and _update_actor
 * (:514-524) with PPOLag's surrogate advantage (adv_r - lambda adv_c)/(1 + lambda)
 * (naive_lagrange/ppo_lag.py:101-102) and loss_kind 0 = PPO clipped loss (base/ppo.py:66-78) or
 * 1 = plain ratio * adv (policy_gradient.py:574-578).  Data arrays are the ENV-MAJOR tensors of
 * osa_buffer_get; idx[B] selects the minibatch rows (NULL = rows 0..B-1); *lagrange is a device scalar.
 * mode 0: gradient + local clip + Adam;  1: gradient + local clip (caller all-reduces grads[3][P]
 * and calls osa_adam_apply -- clip-then-average order of policy_gradient.py:437-442);  2: raw grads.
 * B <= 64*max_blocks rows are processed by ceil(B/64) workgroups per network (ws: at least
 * osa_minibatch_ws_floats floats when more than one is used, ZERO-INITIALISED ONCE by the caller: its tail
 * holds the partial norms and the arrival / finish counters of the fused slab-reduce + clip/Adam launch, which
 * every call leaves re-armed at zero; use one ws with ONE value of max_blocks).
 * step_stats[16] receives
 *   [0] mse_r [1] mse_c [2] loss_pi [3] mean ratio [4] entropy [5] sum p^2 (V_r) [6] sum p^2 (V_c)
 *   [7] |g_pi| [8] |g_Vr| [9] |g_Vc|   (logged Loss_*_critic = mse + critic_norm_coef * sum p^2). */
size_t osa_minibatch_ws_floats(int obs_dim, int act_dim, int hidden, int max_blocks);
int osa_ppo_minibatch(int obs_dim, int act_dim, int hidden, float* params, float* adam_m,
                      float* adam_v, int* adam_step, float* grads, const float* obs, int ld_obs,
                      const float* act, int ld_act, const float* logp, const float* target_value_r,
                      const float* target_value_c, const float* adv_r, const float* adv_c,
                      const long* idx, int B, const float* lagrange, const osa_ppo_hparams* hp,
                      int loss_kind, int mode, int nets_mask, int max_blocks, float* ws,
                      float* step_stats, void* stream);

/* Extended actor surrogates for the sibling first-order algorithms; everything else is
 * osa_ppo_minibatch.  Per-sample actor loss
 *     l_i = mbar * ratio_scale * s_i + mask_i * kl_coef * KL(pi_theta(.|s_i) || pi_old(.|s_i)),
 * s_i the loss_kind surrogate, mask_i = 1[KL_i <= kl_mask_eta], mbar = mean_i(mask_i) (kl_mask_eta < 0:
 * mask_i = mbar = 1) -- exactly what focops.py:84-88 differentiates (its (B,1) x (B,) broadcast makes the
 * surrogate term see the minibatch MEAN of the mask) -- plus
 * cost_kappa * relu(mean_i(ratio_i * adv_c_i) + cost_excess) when cost_kappa > 0 (value in step_stats[10]).
 * mbar and the penalty are minibatch-level quantities: with a mask or a penalty B <= 64 is required
 * (one block; OSA_EUNSUPPORTED otherwise).
 *   FOCOPS (first_order/focops.py:83-92): loss_kind 1, kl_coef 1, kl_mask_eta = focops_eta,
 *                                         ratio_scale = 1/focops_lam
 *   CUP second stage (first_order/cup.py:96-103): loss_kind 1 on adv = -lambda*coef*adv_c, kl_coef 1
 *   P3O (penalty_function/p3o.py:62-68,112-114): loss_kind 0 + cost_kappa = kappa, cost_excess = Jc - limit
 * old_mean[rows][ld_old_mean] / old_log_std[act_dim]: the behaviour policy's Gaussian per sample (same row
 * indexing as obs).  ext == NULL: identical to osa_ppo_minibatch. */
typedef struct osa_surrogate_ext {
  const float* old_mean;
  int ld_old_mean;
  const float* old_log_std;
  float kl_coef;
  float kl_mask_eta;
  float ratio_scale;
  float cost_kappa;
  float cost_excess;
} osa_surrogate_ext;
int osa_ppo_minibatch_ext(int obs_dim, int act_dim, int hidden, float* params, float* adam_m,
                          float* adam_v, int* adam_step, float* grads, const float* obs, int ld_obs,
                          const float* act, int ld_act, const float* logp,
                          const float* target_value_r, const float* target_value_c,
                          const float* adv_r, const float* adv_c, const long* idx, int B,
                          const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int mode,
                          int nets_mask, int max_blocks, float* ws, float* step_stats,
                          const osa_surrogate_ext* ext, void* stream);

/* Persistent form of the same update: ONE launch runs a whole pass of PolicyGradient._update's inner
 * loop (policy_gradient.py:366-382) -- ceil(M/B) dependent minibatch steps over the permutation
 * perm[M] (NULL = identity) -- for all three networks, with the parameters resident in LDS
 * and the Adam moments in registers for the whole pass (see csrc/ppo_pass_kernel.hip).  Same
 * arithmetic per step as osa_ppo_minibatch(mode 0); step_stats[ceil(M/B)][16] as above (columns 10..15
 * additionally receive the three networks' Adam bias-correction factors of the step).  A step with
 * B > 64 rows accumulates ceil(B/64) chunks in the accumulator registers before its clip + Adam.  Single
 * process only (world_size == 1: the data-parallel path needs an all-reduce between gradient and
 * Adam and uses osa_ppo_minibatch).  osa_ppo_pass_supported: 1 if (obs_dim, act_dim, hidden) fit.
 * The observation rows must be 16-byte aligned with ld_obs % 4 == 0 (OSA_EUNSUPPORTED otherwise -- pad
 * the rows; the same holds for osa_ppo_dp_step / osa_ppo_dp_pass). */
int osa_ppo_pass_supported(int obs_dim, int act_dim, int hidden);
int osa_ppo_pass(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                 int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                 const float* logp, const float* target_value_r, const float* target_value_c,
                 const float* adv_r, const float* adv_c, const long* perm, long M, int B,
                 const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                 float* step_stats, void* stream);
/* The same persistent pass for WIDE observations (65 <= obs_dim <= 512; callers prefer osa_ppo_pass where it
 * is supported; e.g. SafetyHumanoidVelocity 376/17;
 * model builder omnisafe/utils/model.py:73-111 takes any sizes): the first layer's weights and Adam moments do
 * not fit one compute unit, so they stay in (L2-resident) global memory -- W1 fragments are streamed by the
 * forward pass, the W1 gradient is accumulated in MFMA accumulator registers and norm + clip + Adam are applied
 * by the lanes that hold it; the other layers live in LDS / registers as in osa_ppo_pass (see
 * csrc/wide_pass_kernel.hip).  B <= 64, loss_kind 0/1, single process; same arguments, statistics and
 * arithmetic per step as osa_ppo_pass / osa_ppo_minibatch(mode 0), plus ws: osa_ppo_wide_pass_ws_floats
 * floats of scratch (a tiled private copy of the first layer, of its Adam moments and of its gradient for the
 * duration of the launch; contents undefined afterwards).  OSA_EUNSUPPORTED otherwise. */
int osa_ppo_wide_pass_supported(int obs_dim, int act_dim, int hidden);
size_t osa_ppo_wide_pass_ws_floats(int obs_dim, int act_dim, int hidden);
int osa_ppo_wide_pass(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                      int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                      const float* logp, const float* target_value_r, const float* target_value_c,
                      const float* adv_r, const float* adv_c, const long* perm, long M, int B,
                      const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                      float* ws, float* step_stats, void* stream);
/* osa_ppo_wide_pass with the first layer SPLIT over cooperating compute units (csrc/wide_split_kernel.hip):
 * a network is 1 + ceil(KB / 6) workgroups of one cooperative launch -- every helper owns <= 96 input columns of
 * W1 (LDS) and their Adam moments (registers) for the whole pass, the leader owns the other layers; three
 * hand-offs per step (partial pre-activations, dz1, squared-norm shares) through `xch`
 * (osa_ppo_split_pass_xch_floats floats, zero-initialised once).  local = 0: the workgroups are spread over the
 * XCCs and xch MUST come from osa_dp_exchange_alloc (uncached device memory; OSA_EINVAL otherwise).  local = 1:
 * the workgroups of a network are placed on ONE XCC (blocks net + 8 role of an 8 (C + 1) grid) and xch MUST be
 * ordinary device memory: the hand-offs are then served by that XCC's L2; the kernel verifies the placement
 * (XCC_ID of every workgroup) BEFORE anything is modified and, if it does not hold, every workgroup returns with
 * parameters, Adam state and step counters untouched and the sticky flag set to 2 (repeat with local = 0).
 * Same arguments, statistics and per-step arithmetic as osa_ppo_wide_pass (float32 re-association of the
 * layer-1 sum only).  OSA_EUNSUPPORTED when the shape is not supported or the device cannot hold the
 * workgroups together: use osa_ppo_wide_pass.  A peer that never arrives raises a sticky flag instead of
 * hanging the device: osa_ppo_split_pass_timed_out(xch, &flag) reads it (synchronous copy; 0 = fine). */
int osa_ppo_split_pass_supported(int obs_dim, int act_dim, int hidden);
size_t osa_ppo_split_pass_xch_floats(int obs_dim, int act_dim, int hidden);
int osa_ppo_split_pass(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                       int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                       const float* logp, const float* target_value_r, const float* target_value_c,
                       const float* adv_r, const float* adv_c, const long* perm, long M, int B,
                       const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                       float* xch, int local, float* step_stats, void* stream);
int osa_ppo_split_pass_timed_out(const float* xch, int* out);
/* Clears the sticky word after a tripped PLACEMENT check (flag 2: nothing was modified, the caller repeats the pass
 * with another placement).  Never clear a time-out (flag 1). */
int osa_ppo_split_pass_clear_flag(float* xch);
/* The DATA-PARALLEL form of osa_ppo_split_pass (world_size > 1 without a per-step cross-GPU collective, as
 * osa_ppo_dp_pass does for narrow observations): the arrays hold the all-gathered rollouts of `world` ranks (rank r:
 * rows r M .. r M + M - 1), perm is [world][M] (row r = rank r's shuffle of its OWN M rows, indices in [0, M)).
 * One cooperative launch of world x 3 x (1 + ceil(KB / 6)) workgroups: every virtual rank is a full set of leader
 * + helpers computing ITS minibatch's gradient and ITS clip factor (avg_grads order of the reference:
 * clip_grad_norm_ locally, then average -- policy_gradient.py:437-442, 478-483, 519-524; utils/distributed.py:167-198);
 * then the `world` owners of the same parameters (helper c of every rank; the leaders) exchange their clipped
 * shares, sum them in rank order, divide by world and apply the SAME Adam step to their own replica: no parameter
 * ever crosses between replicas, they stay bit-identical, rank 0's is written back.  step_stats receive the
 * rank-averaged statistics (what Logger.get_stats averages).  xch: osa_ppo_split_dp_xch_floats(...) floats from
 * osa_dp_exchange_alloc (uncached; OSA_EINVAL otherwise), zero-initialised once; its sticky word (1: a workgroup
 * never arrived, 2: placement) is read with osa_ppo_split_pass_timed_out.
 * place = 0, dpx = NULL: rank-major workgroups spread over the XCCs, every hand-off through xch.
 * place = 1: the `world` owners of the same parameters sit on ONE XCC (owner group g = (network, role) on XCC
 * g mod 8) and average through dpx = osa_ppo_split_dp_dpx_floats(...) floats of ORDINARY device memory (zeroed
 * once; OSA_EINVAL for an uncached buffer) served by that XCC's L2, the intra-rank hand-offs stay in xch; the
 * kernel verifies the placement before it modifies anything and otherwise returns untouched with the sticky word
 * at 2 (repeat with place = 0).  OSA_EUNSUPPORTED when the device cannot hold the workgroups together
 * (world x 3 x (C + 1) > compute units; place = 1: ceil(3 (C + 1) / 8) x world > compute units / 8) or the shape
 * is outside osa_ppo_split_pass_supported: use the per-step path (osa_ppo_minibatch mode 1 + all-reduce +
 * osa_adam_apply). */
size_t osa_ppo_split_dp_xch_floats(int obs_dim, int act_dim, int hidden, int world);
size_t osa_ppo_split_dp_dpx_floats(int obs_dim, int act_dim, int hidden, int world);
int osa_ppo_split_dp_pass(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                          int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                          const float* logp, const float* target_value_r, const float* target_value_c,
                          const float* adv_r, const float* adv_c, const long* perm, long M, int B, int world,
                          const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                          float* xch, float* dpx, int place, float* step_stats, void* stream);
/* osa_ppo_pass with the extended actor surrogates of osa_ppo_minibatch_ext (FOCOPS, CUP's second stage,
 * P3O): B <= 64 (the trust-mask mean and the penalty are minibatch-level quantities of one 64-row block);
 * OSA_EUNSUPPORTED otherwise -- use osa_ppo_minibatch_ext.  ext == NULL: osa_ppo_pass.  With cost_kappa > 0
 * step_stats[10] of every step receives the penalty value. */
int osa_ppo_pass_ext(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                 int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                 const float* logp, const float* target_value_r, const float* target_value_c,
                 const float* adv_r, const float* adv_c, const long* perm, long M, int B,
                 const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                 float* step_stats, const osa_surrogate_ext* ext, void* stream);

/* Data-parallel optimiser step WITHOUT a per-step cross-GPU collective ("replicated data").  The data
 * arrays hold the all-gathered env-major rollouts of all `world` ranks ([world * M] rows, rank r at
 * rows r*M ..), perm[world][M] each rank's own permutation (local row indices).  One call performs
 * optimiser step `step_index` of the pass for every virtual rank -- `world` workgroups per network
 * compute the locally clipped gradients of rank r's minibatch perm[r][step*B ..] -- then averages them
 * (distributed.avg_grads, omnisafe/utils/distributed.py:193-198, clip-then-average order of
 * policy_gradient.py:437-442) and applies Adam.  Every rank executes the same arithmetic on the same
 * data, so replicas stay bit-identical without exchanging gradients or parameters.  lr_dev: optional
 * device float[2] {lr_actor, lr_critic} overriding hp (keeps a captured hipGraph valid across epochs).
 * slabs: osa_ppo_dp_ws_floats(...) floats.  step_stats[16]: rank-averaged statistics.  The Adam step
 * count used is adam_step[net] + step_index + 1; call osa_ppo_dp_end_pass once after the last step of
 * a pass to advance adam_step by the number of steps. */
size_t osa_ppo_dp_ws_floats(int obs_dim, int act_dim, int hidden, int world);
int osa_ppo_dp_step(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                    int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                    const float* logp, const float* target_value_r, const float* target_value_c,
                    const float* adv_r, const float* adv_c, const long* perm, long M, int B, int world,
                    int step_index, const float* lagrange, const osa_ppo_hparams* hp,
                    const float* lr_dev, int loss_kind, int nets_mask, float* slabs,
                    float* step_stats, void* stream);
int osa_ppo_dp_end_pass(int* adam_step, int nets_mask, int nsteps, void* stream);
/* The two halves of osa_ppo_dp_step for the PER-STEP ALL-REDUCE mode over real ranks (round 5) -- the reference's own
 * structure, policy_gradient.py:437-443 with distributed.py:167-198: clip_grad_norm_, avg_grads, optimizer.step -- on
 * the persistent pass kernel's gradient-only form instead of the per-step kernels (weights staged in LDS once per
 * launch, gradients from registers: 13 us against 48 per 64-row step).
 *   phase 1: the locally clipped gradients and statistics of THIS rank's minibatch (world = 1, perm = its row indices,
 *            M rows of them, step_index = 0) into slabs[3][P + 16]; nothing else is touched;
 *   [caller: ONE flat all-reduce (average) of the slabs]
 *   phase 2: Adam from the slabs (step count adam_step[net] + step_index + 1: pass the step's index within the pass
 *            and call osa_ppo_dp_end_pass after its last step); only the parameter / moment / slab / hp arguments are used.
 *   phase 0: osa_ppo_dp_step. */
int osa_ppo_dp_step_phase(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                          int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                          const float* logp, const float* target_value_r, const float* target_value_c,
                          const float* adv_r, const float* adv_c, const long* perm, long M, int B, int world,
                          int step_index, const float* lagrange, const osa_ppo_hparams* hp,
                          const float* lr_dev, int loss_kind, int nets_mask, float* slabs,
                          float* step_stats, int phase, void* stream);

/* The same replicated-data optimiser chain as ONE cooperative persistent launch per pass: 3 x world
 * workgroups stay resident for all ceil(M/B) steps; workgroup (net, r) keeps its own LDS/register copy
 * of the network and of the Adam moments, computes rank r's locally clipped gradient, publishes it in
 * `exchange` (osa_ppo_dp_pass_ws_floats floats), meets its `world` peers at an agent-scope arrival
 * counter (sync: int[4]: sync[0..2] arrival counters, reset by every call; sync[3] STICKY time-out flag,
 * never reset by the library -- the caller zeroes all four words once at allocation and reads sync[3]
 * at its next host synchronisation; a peer that never arrives is flagged after a bounded spin instead of
 * hanging the device), then sums the
 * gradients in rank order and applies Adam locally -- every peer does identical arithmetic, rank 0's
 * copy is written back at the end and adam_step advances by the number of steps.  Launched with
 * hipLaunchCooperativeKernel, i.e. the runtime verifies that all 3 * world workgroups are co-resident
 * (OSA_EUNSUPPORTED otherwise: use osa_ppo_dp_step).
 * Replaces: the minibatch loop of PolicyGradient._update under torch.distributed
 * (policy_gradient.py:366-382, 437-442; distributed.py:193-198). */
/* Optional: the exchange buffer of osa_ppo_dp_pass in UNCACHED device memory (hipExtMallocWithFlags,
 * hipDeviceMallocUncached).  The kernel recognises such a buffer and hands the gradients over without
 * the agent-scope L2 write-back / invalidate that ordinary (cached) device memory needs.  The one place
 * where the library allocates: the memory type cannot be requested through torch.  OSA_EUNSUPPORTED if
 * the runtime refuses the flag (use an ordinary buffer then). */
int osa_dp_exchange_alloc(size_t floats, float** out);
int osa_dp_exchange_free(float* p);
size_t osa_ppo_dp_pass_ws_floats(int obs_dim, int act_dim, int hidden, int world);
int osa_ppo_dp_pass(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                    int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                    const float* logp, const float* target_value_r, const float* target_value_c,
                    const float* adv_r, const float* adv_c, const long* perm, long M, int B, int world,
                    const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                    float* exchange, int* sync, float* step_stats, void* stream);
/* osa_ppo_dp_pass with an explicit placement.  local = 0: as osa_ppo_dp_pass (grid 3 x world, spread over the
 * XCCs; `exchange` uncached or ordinary memory).  local = 1: the `world` workgroups of a network are the blocks
 * net + 8 r of an 8 x world grid, i.e. they run on ONE XCC, and `exchange` MUST be ordinary device memory
 * (OSA_EINVAL for an osa_dp_exchange_alloc buffer): the hand-offs are served by that XCC's L2.  sync: int[8]
 * (zeroed once by the caller): sync[4..6] receive the bit masks of the XCCs each network's workgroups ran on,
 * sync[7] counts arrivals.  The placement is verified BEFORE anything is modified: if a network's mask holds
 * more than one bit (or a workgroup never arrives) EVERY workgroup returns with parameters, Adam state and step
 * counters untouched and the sticky flag sync[3] is 2 (1): repeat the call with local = 0.  world <= CUs / 8. */
int osa_ppo_dp_pass_placed(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                    int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                    const float* logp, const float* target_value_r, const float* target_value_c,
                    const float* adv_r, const float* adv_c, const long* perm, long M, int B, int world,
                    const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                    float* exchange, int* sync, int local, float* step_stats, void* stream);

/* ONE-SHOT PEER EXCHANGE (round 6; csrc/p2p_pass_kernel.hip): the data-parallel minibatch loop with REAL ranks and no
 * collective on the step path -- SURVEY.md 8(e) "xGMI mapping" (one-shot all-to-all write of the <= 336 KB clipped
 * gradients + local reduce).  Replaces per optimiser step `clip_grad_norm_` -> `distributed.avg_grads` (19 blocking
 * all-reduces) -> `optimizer.step()`: omnisafe/algorithms/on_policy/base/policy_gradient.py:437-443, 478-483, 519-524;
 * omnisafe/utils/distributed.py:167-198.
 *
 * Every rank (one process per GPU) owns ONE exchange buffer of osa_p2p_exchange_floats(...) floats in UNCACHED device
 * memory: osa_p2p_exchange_alloc allocates it zeroed and returns its 64-byte IPC handle (hipIpcGetMemHandle); the host
 * side exchanges the handles (torch.distributed all_gather: a set-up step, once) and maps every peer's buffer with
 * osa_p2p_exchange_open (hipIpcOpenMemHandle; peers may be other devices of the node or other processes on the same
 * device).  osa_p2p_exchange_release unmaps a peer's buffer / frees an own one.  All ranks must have opened all
 * buffers (a barrier) before the first pass.
 *
 * osa_ppo_p2p_pass = osa_ppo_pass on THIS rank's rows (arrays, perm, M, B as there: 3 workgroups, weights in LDS, Adam
 * moments in registers, all ceil(M / B) steps in one launch), except that after the local clip each network's
 * workgroup writes its gradient slab into the buffer of every rank (peers[q], q = 0 .. world - 1 as mapped into this
 * process; peers[rank] = the own buffer), releases at system scope, stores the step's sequence number into its arrival
 * word in every buffer, waits for the `world` arrival words of its own buffer, adds the slabs IN RANK ORDER, divides
 * by world (clip-then-average) and applies Adam: identical arithmetic on every rank, bit-identical replicas, every
 * rank writes its own parameters / moments back.  step_stats: rank-averaged (what Logger.get_stats averages).
 * seq0: optimiser steps exchanged through these buffers before this call -- the SAME on every rank; the caller adds
 * ceil(M / B) after each call (arrival words only grow: nothing is reset between launches).  timeout_s: how long a
 * workgroup waits for a peer before it sets the sticky word read by osa_p2p_exchange_timed_out (the pass then
 * finishes without waiting: results invalid, the device never hangs).  world <= 16; B <= 2048 (B > 64: the workgroup
 * walks through the minibatch's 64-row chunks); shapes as osa_ppo_pass_supported (OSA_EUNSUPPORTED otherwise). */
size_t osa_p2p_exchange_floats(int obs_dim, int act_dim, int hidden, int world);
int osa_p2p_exchange_alloc(size_t floats, float** out, void* ipc_handle64);
int osa_p2p_exchange_open(const void* ipc_handle64, float** out);
int osa_p2p_exchange_release(float* p);
int osa_p2p_exchange_timed_out(const float* own, int* flag);
int osa_ppo_p2p_pass(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                     int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                     const float* logp, const float* target_value_r, const float* target_value_c,
                     const float* adv_r, const float* adv_c, const long* perm, long M, int B, int world, int rank,
                     float* const* peers, unsigned seq0, double timeout_s, const float* lagrange,
                     const osa_ppo_hparams* hp, int loss_kind, int nets_mask, float* step_stats, void* stream);

/* Single-process persistent pass for minibatches of 64 < B <= 2048 rows (the trust-region family's critic
 * updates: batch_size 128, natural_pg.py:205-223) with the ceil(B / 64) 64-row CHUNKS of a minibatch on cooperating
 * workgroups (one compute unit each, per network) instead of one workgroup walking through them: every workgroup
 * keeps the network and its Adam state, computes the raw gradient of its chunk scaled by 1 / rows of the minibatch,
 * the chunks' gradients are summed (exchange: osa_ppo_dp_pass_ws_floats(.., world = ceil(B / 64)) floats, zeroed
 * once), the SUM is clipped by its norm and everybody applies the same Adam step: the arithmetic of
 * osa_ppo_pass / osa_ppo_minibatch(mode 0) on the same rows (float32 re-association of the sum over chunks only).
 * Arguments as osa_ppo_pass plus exchange / sync / local as in osa_ppo_dp_pass_placed (sync: int[8]; sticky
 * sync[3]).  OSA_EUNSUPPORTED for B <= 64 (use osa_ppo_pass) or when the workgroups cannot be co-resident. */
int osa_ppo_chunked_pass(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                         int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                         const float* logp, const float* target_value_r, const float* target_value_c,
                         const float* adv_r, const float* adv_c, const long* perm, long M, int B,
                         const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                         float* exchange, int* sync, int local, float* step_stats, void* stream);
/* osa_ppo_chunked_pass UNDER DATA PARALLELISM (the critic passes of the trust-region family, batch 128, with
 * world_size > 1 -- BASELINE configs 3 and 5; natural_pg.py:205-223 + distributed.py:167-198): arrays and perm as
 * osa_ppo_dp_pass ([world][M] rows, rank r's shuffle of its own rows), world x ceil(B / 64) workgroups per network
 * in one cooperative launch.  Two hand-offs per optimiser step: inside a rank's chunk group (sum of the chunks'
 * gradients, ITS norm, the rank's clip factor -- the arithmetic of one B-row step of that rank), then across ranks
 * (every chunk workgroup publishes its share of the tiles of the rank's sum; everybody adds the `world` clipped sums
 * in rank order, / world) and the same Adam step in every replica.  exchange: osa_ppo_dp_chunked_pass_ws_floats
 * floats (placement rules as osa_ppo_dp_pass_placed); sync: int[64], zero before the first call ([3] sticky
 * time-out / placement flag as there). */
size_t osa_ppo_dp_chunked_pass_ws_floats(int obs_dim, int act_dim, int hidden, int B, int world);
int osa_ppo_dp_chunked_pass(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                            int* adam_step, const float* obs, int ld_obs, const float* act, int ld_act,
                            const float* logp, const float* target_value_r, const float* target_value_c,
                            const float* adv_r, const float* adv_c, const long* perm, long M, int B, int world,
                            const float* lagrange, const osa_ppo_hparams* hp, int loss_kind, int nets_mask,
                            float* exchange, int* sync, int local, float* step_stats, void* stream);

/* Debugging aid: when set to a device buffer of 48 int64, every osa_ppo_minibatch launch records
 * s_memtime phase timestamps [3 networks][16] (used by tools/phase_clocks.py); NULL disables it. */
int osa_debug_set_clock_buffer(long long* dev_ptr);
/* same for osa_ppo_pass: accumulated cycles per phase over the pass, [3 networks][16]. */
int osa_debug_set_pass_clock_buffer(long long* dev_ptr);
/* Debug: per-workgroup {start, end} clocks of the balanced partial-gradient launch of the large-batch step
 * (long long[16 x workgroups]: 100 MHz constant clock start, shader cycles start, the same two at the end; [4..6] shader
 * cycles after the prologue / the first chunk / the last chunk in the -DOSA_PART_CLOCKS build); NULL = off. */
int osa_debug_set_part_clock_buffer(long long* dev_ptr);

/* ---- General actor-critic MLPs (csrc/general_mlp.hip, round 4): ANY hidden_sizes the reference's model builder
 * accepts (omnisafe/utils/model.py:73-111 build_mlp_network: any depth, any widths; actor and critics configured
 * independently, models/actor_critic/actor_critic.py:60-136), e.g. the 1024 x 1024 networks of the reference's
 * published timing table (docs/source/start/efficiency.rst:15-23).  Layer-wise on one float32-MFMA GEMM kernel
 * (forward / backward-data / backward-weight with fused epilogues); the entry points mirror the per-step family
 * above (osa_policy_step_scaled, osa_ppo_minibatch, osa_adam_apply, osa_actor_kl, osa_actor_eval,
 * osa_actor_fvp_raw) with the network shapes in a descriptor instead of the `hidden` word.
 * Parameter block of one network ([3][P] floats as everywhere): per linear layer l  W_l [out_l][ld_l] (ld_l = in_l
 * rounded up to 4, padding zero) | b_l [out_l rounded up to 4]; the actor's log_std [act_dim] follows its last layer.
 * osa_gmlp_layout fills out[0] = P, out[1] = offset of log_std, then {oW, ob, ld} for network 0..2 x layer
 * 0..OSA_GMLP_MAX_LAYERS-1 (-1 / 0 for absent layers): 2 + 72 ints.
 * ws: osa_gmlp_ws_floats(desc, rows) floats of scratch for a call over `rows` rows (layer outputs, dL/dz, partial
 * gradient slabs); contents irrelevant between calls. */
#define OSA_GMLP_MAX_LAYERS 8
typedef struct osa_gmlp_desc {
  int obs_dim, act_dim;
  int n_layers[3];                    /* linear layers (hidden layers + 1) of actor, reward critic, cost critic */
  int width[3][OSA_GMLP_MAX_LAYERS];  /* output width of every linear layer; the last one = act_dim (actor) / 1 */
  int activation[3];                  /* hidden activation: 0 tanh 1 relu 2 sigmoid 3 softplus 4 identity */
} osa_gmlp_desc;
int osa_gmlp_layout(const osa_gmlp_desc* desc, int* out);
size_t osa_gmlp_ws_floats(const osa_gmlp_desc* desc, long rows);
/* ConstraintActorCritic.step (constraint_actor_critic.py:84-109) + ActionScale: arguments as osa_policy_step_scaled. */
int osa_gmlp_policy_step(const osa_gmlp_desc* desc, const float* params, const float* obs, int ld_obs, long N,
                         const float* eps, unsigned long long seed, unsigned long long offset,
                         const unsigned long long* offset_base, int deterministic, int nets_mask, float* act,
                         int ld_act, float* value_r, float* value_c, float* logp, float* mean_out, int ld_mean,
                         float* act_env, int ld_env, const float* old_min, const float* old_max, float min_action,
                         float max_action, float* ws, size_t ws_floats, void* stream);
/* One optimiser step of PolicyGradient._update's inner loop (policy_gradient.py:428-445, 468-485, 514-524): arguments
 * and modes as osa_ppo_minibatch (mode 0 grad + local clip + Adam, 1 grad + local clip -> grads[3][P], 2 raw grads).
 * loss_kind 2 = the raw Fisher-vector product of the actor along `vec` (padded actor vector; fvp_scale = 1 / (M D_a);
 * natural_pg.py:91-119) into grads[0], as osa_actor_fvp_raw. */
int osa_gmlp_minibatch(const osa_gmlp_desc* desc, float* params, float* adam_m, float* adam_v, int* adam_step,
                       float* grads, const float* obs, int ld_obs, const float* act, int ld_act, const float* logp,
                       const float* target_value_r, const float* target_value_c, const float* adv_r,
                       const float* adv_c, const long* idx, long B, const float* lagrange, const osa_ppo_hparams* hp,
                       int loss_kind, int mode, int nets_mask, const float* vec, float fvp_scale, float* ws,
                       size_t ws_floats, float* step_stats, void* stream);
/* osa_gmlp_minibatch with the extended actor surrogates of osa_ppo_minibatch_ext (FOCOPS first_order/focops.py:83-92,
 * CUP's second stage first_order/cup.py:96-103, P3O penalty_function/p3o.py:62-68,112-114) on general networks -- the
 * reference builds any hidden_sizes for them as for every other algorithm (utils/model.py:73-111).  ext == NULL:
 * osa_gmlp_minibatch.  The trust-mask mean and the penalty are minibatch-level quantities: with kl_mask_eta >= 0 or
 * cost_kappa > 0 at most 256 rows (OSA_EUNSUPPORTED otherwise); step_stats[10] receives the penalty value.
 * Minibatches of at most 64 rows (the YAML batch_size) run on the skinny kernels (weights streamed once per pass,
 * clip + Adam from recomputed gradient tiles; OSA_GMLP_SKINNY=0: the tiled GEMM path). */
int osa_gmlp_minibatch_ext(const osa_gmlp_desc* desc, float* params, float* adam_m, float* adam_v, int* adam_step,
                           float* grads, const float* obs, int ld_obs, const float* act, int ld_act, const float* logp,
                           const float* target_value_r, const float* target_value_c, const float* adv_r,
                           const float* adv_c, const long* idx, long B, const float* lagrange,
                           const osa_ppo_hparams* hp, int loss_kind, int mode, int nets_mask, const float* vec,
                           float fvp_scale, float* ws, size_t ws_floats, float* step_stats,
                           const osa_surrogate_ext* ext, void* stream);
/* osa_adam_apply for general networks; fin8x3: 24 floats of device scratch. */
int osa_gmlp_adam_apply(const osa_gmlp_desc* desc, float* params, float* adam_m, float* adam_v, int* adam_step,
                        float* grads, const osa_ppo_hparams* hp, int nets_mask, float* fin8x3, void* stream);
/* osa_actor_kl (kind 0: old_mean == NULL snapshots the mean into mean_out; otherwise KL(old || new) -> out[0] with
 * reduce_mode as there) and osa_actor_eval (kind 1: out[0..3] = loss_pi, loss_cost, KL, mean ratio) for general
 * networks.  actor_params: the actor's block (network 0 of a [3][P] tensor, or a candidate vector). */
int osa_gmlp_actor_stats(const osa_gmlp_desc* desc, const float* actor_params, const float* obs, int ld_obs, long M,
                         const float* old_mean, int ld_old, const float* old_log_std, int kind, int reduce_mode,
                         const float* act, int ld_act, const float* logp, const float* adv_r, const float* adv_c,
                         const float* lagrange, float* mean_out, int ld_mean, float* ws, size_t ws_floats, float* out,
                         void* stream);

/* Adam step on already clipped (and, for world_size > 1, all-reduce-averaged) gradients. */
int osa_adam_apply(int obs_dim, int act_dim, int hidden, float* params, float* adam_m, float* adam_v,
                   int* adam_step, float* grads, const osa_ppo_hparams* hp, int nets_mask,
                   void* stream);

/* Full-batch actor forward over M env-major rows.  With old_mean == NULL it only snapshots the
 * distribution mean into mean_out (old_distribution = actor(obs), policy_gradient.py:357).  Otherwise
 * it evaluates KL(old || new) of torch.distributions.kl._kl_normal_normal: reduce_mode 0 =
 * .sum(-1).mean() (policy_gradient.py:383-389), 1 = .mean() over all M x act_dim elements
 * (natural_pg.py:95, trpo.py:113, cpo.py:133); the scalar lands in *kl_out (device).
 * ws: at least 1024 doubles. */
int osa_actor_kl(int obs_dim, int act_dim, int hidden, const float* actor_params, const float* obs,
                 int ld_obs, long M, const float* old_mean, int ld_old, const float* old_log_std,
                 int reduce_mode, float* mean_out, int ld_mean, double* ws, float* kl_out,
                 void* stream);

/* ------------------------------------------------------------------------------------------------
 * Rollout step (replaces the wrapper chain and the per-env loop of OnPolicyAdapter.rollout)
 * ---------------------------------------------------------------------------------------------- */

/* Normalizer._push (omnisafe/common/normalizer.py:109-139) with a batch of the rows of x selected by
 * mask (NULL = all N rows; zero selected rows = no-op): Chan/Golub/LeVeque merge of the batch mean and
 * centred sum of squares into the running state, var = sumsq/(count-1), std = max(sqrt(var), 1e-2).
 * State: mean/sumsq/var/std float32[D], count int64[1] (all device).  ws: osa_normalizer_ws_doubles(N, D)
 * doubles, ZERO-INITIALISED ONCE by the caller (its FIRST word is the arrival ticket of the single-launch
 * reduction; every call leaves it at zero again, so one workspace sized for the largest N serves any
 * smaller batch afterwards). */
size_t osa_normalizer_ws_doubles(int N, int D);
int osa_normalizer_push(const float* x, int ld, int N, int D, const uint8_t* mask, float* mean,
                        float* sumsq, float* var, float* std_, long* count, double* ws,
                        void* stream);

/* Normalizer.normalize's output (normalizer.py:102-107): y = clamp((x - mean)/std, -clip, clip) for
 * rows selected by mask (others copied unchanged); x copied unchanged while count <= 1. */
int osa_normalizer_apply(const float* x, int ld_x, float* y, int ld_y, int N, int D,
                         const uint8_t* mask, const float* mean, const float* std_, const long* count,
                         float clip, void* stream);

/* ActionScale.step (omnisafe/envs/wrapper.py:510-514):
 * out = old_min + (old_max - old_min) * (act - min_action) / (max_action - min_action). */
int osa_action_scale(const float* act, int ld_act, float* out, int ld_out, int N, int act_dim,
                     const float* old_min, const float* old_max, float min_action, float max_action,
                     void* stream);

/* The per-env loop of OnPolicyAdapter.rollout (omnisafe/adapter/onpolicy_adapter.py:86-136) for all N
 * envs at one step: episode return/cost/length accumulation (_log_value :155-157), path-end flag and
 * bootstrap selection -- 0 if terminated, V(final_observation) (vfinal_*) if truncated, V(next_obs)
 * (vnext_*) at epoch end (:114-126) -- written into row t of the buffer's path_end/boot_r/boot_c, and
 * the finished episodes' metrics (_log_metrics :159-174) into row t of ep_done/ep_*_out, after which
 * the per-env accumulators reset (:128-134).  vnext_* / vfinal_* may be NULL when not needed.
 * reward_row / cost_row (each may be NULL): row t of the buffer's reward / cost columns, filled with the step's
 * reward / cost (buffer.store, :100-108) in the same launch -- for callers without a reward / cost normaliser or an
 * adapter hook that rewrites the row in between (round 3: two copy launches per vector step less). */
int osa_rollout_post_step(int N, int epoch_end, const float* reward, const float* cost,
                          const uint8_t* terminated, const uint8_t* truncated, const float* vnext_r,
                          const float* vnext_c, const float* vfinal_r, const float* vfinal_c,
                          float* ep_ret, float* ep_cost, float* ep_len, uint8_t* path_end,
                          float* boot_r, float* boot_c, uint8_t* ep_done, float* ep_ret_out,
                          float* ep_cost_out, float* ep_len_out, float* reward_row, float* cost_row,
                          void* stream);

/* End-of-epoch log flush of OnPolicyAdapter.rollout (_log_metrics onpolicy_adapter.py:159-174; logger.store of the
 * value means :88-92) in two launches: the finished episodes (done[i] != 0, i = t N + n over the epoch's M = T N
 * slots) compacted in (step, env) order -- out_idx[k] = i, out_vals[0 M + k] = ep_ret[i], [1 M + k] = ep_cost[i],
 * [2 M + k] = ep_len[i], [3 M + k] = extra[i] (extra may be NULL) -- their number in out_count[0], and out_means =
 * { mean(value_r), mean(value_c) } over the M slots (float64 sums in a fixed order).  out_idx: int[M], out_vals:
 * float[4 M]; ws: osa_episode_flush_ws_doubles(M) doubles, zeroed once (every call leaves its ticket at 0). */
size_t osa_episode_flush_ws_doubles(long M);
int osa_episode_flush(const uint8_t* done, const float* ep_ret, const float* ep_cost, const float* ep_len,
                      const float* extra, long M, const float* value_r, const float* value_c, int* out_count,
                      int* out_idx, float* out_vals, float* out_means, double* ws, void* stream);

/* *out = mean of x[idx[0 .. n)] (idx NULL: of x[0 .. n)), float64 accumulation in a fixed order: the Value/Adv the
 * reference logs from the LAST minibatch of the update (policy_gradient.py:369-377, 402). */
int osa_gather_mean(const float* x, const long* idx, long n, float* out, void* stream);

/* SauteAdapter.step for all N envs (omnisafe/adapter/saute_adapter.py:124-196; SimmerAdapter shares it with
 * reset_value = relative budget): safety_obs <- (safety_obs - cost/budget)/saute_gamma; reward_out =
 * reward if safety_obs > 0 else unsafe_reward; finished envs (terminated | truncated) restart from
 * reset_value; the new safety_obs becomes column `col` of the (already normalised) next-observation rows
 * and, when given, of the final-observation rows; ep_budget accumulates it and a finished episode writes
 * its sum to ep_budget_out (Metrics/EpBudget) and restarts from 0. */
int osa_saute_step(int N, const float* cost, const float* reward, const uint8_t* terminated,
                   const uint8_t* truncated, float* safety_obs, const float* budget, float saute_gamma,
                   float unsafe_reward, const float* reset_value, float* reward_out, float* next_rows,
                   int ld_next, float* final_rows, int ld_final, int col, float* ep_budget,
                   float* ep_budget_out, void* stream);

/* Synthetic fixed-shape vector CMDP for throughput runs (stand-in for Safety-Gymnasium, whose physics
 * is third-party CPU code outside the reference repo; same role as tests/simple_env.py:30-90 of the
 * reference): obs ~ N(0,1)^obs_dim, reward ~ N(0,1), cost ~ Bernoulli(cost_p), never terminates,
 * truncates every `horizon` steps with gymnasium's vector auto-reset convention (the returned obs is
 * the post-reset obs; the pre-reset obs goes to final_obs).  reset_only != 0 draws initial
 * observations and zeroes the step counters.  Random draws are keyed by (seed, step + *step_base, env, k);
 * step_base: device pointer or NULL (same role as osa_policy_step's offset_base: graph replay). */
int osa_synth_env_step(unsigned long long seed, unsigned long long step,
                       const unsigned long long* step_base, int N, int obs_dim,
                       int horizon, float cost_p, int* steps, float* obs, int ld_obs, float* reward,
                       float* cost, uint8_t* terminated, uint8_t* truncated, float* final_obs,
                       int ld_final, int reset_only, void* stream);

/* Learnable synthetic vector CMDP "SynthReach-v0" (obs_dim >= 6, 2 actions; stand-in for a
 * Safety-Gymnasium goal task, which is third-party CPU physics outside the reference repo; plays the
 * role of the reference's tests/simple_env.py:30-90 for learning-curve comparisons): a point moves by
 * 0.1*clip(action,-1,1) inside [-1.5,1.5]^2, reward = decrease of the distance to the goal, +1 and a
 * new goal when within 0.15 of it, cost = 1 inside a disc of radius 0.3 around a hazard; never
 * terminates, truncates every `horizon` steps (gymnasium vector auto-reset convention as above).
 * state: N x 8 floats (p, goal, hazard, 2 pad), owned by the caller, updated in place.
 * obs row = [p, goal-p, hazard-p, 0...].  reset_only != 0 draws fresh states and zeroes `steps`. */
int osa_reach_env_step(unsigned long long seed, unsigned long long step,
                       const unsigned long long* step_base, int N, int obs_dim,
                       int horizon, float* state, int* steps, const float* action, int ld_action,
                       float* obs, int ld_obs, float* reward, float* cost, uint8_t* terminated,
                       uint8_t* truncated, float* final_obs, int ld_final, int reset_only,
                       void* stream);

/* ------------------------------------------------------------------------------------------------
 * Trust-region machinery (NaturalPG / TRPO / TRPOLag / CPO actor update)
 *
 * Flat vectors here are PADDED actor parameter vectors of P floats (osa_mlp_layout; padding entries
 * are zero in every vector, so dot products equal those of the reference's compact flat vectors of
 * omnisafe/utils/tools.py:35-129).  The full-batch policy gradient is osa_ppo_minibatch with idx = NULL,
 * B = M, loss_kind = 1, mode = 2, nets_mask = 1 (raw gradient of -mean(ratio * adv) in grads[0]).
 * ---------------------------------------------------------------------------------------------- */

/* NaturalPG._fvp without damping (omnisafe/algorithms/on_policy/base/natural_pg.py:91-111): the
 * Hessian-vector product of mean KL(pi_old || pi_theta) -- `.mean()` over all M x act_dim elements --
 * at theta = theta_old, restricted to the mean-network parameters: J^T diag(1/sigma^2) J vec / (M D_a),
 * computed as forward-mode JVP followed by the ordinary backward pass (no double backward; exact for
 * a Gaussian policy with state-independent log_std because dKL/dmu = 0 at theta_old).  Result (raw,
 * local to this rank) in grads[0 .. P).  ws: osa_minibatch_ws_floats(.., max_blocks) floats.
 * Hidden width 64 with tanh and observations up to 80 wide run on osa_fvp_kernel (csrc/fvp_kernel.hip: theta and vec
 * resident in LDS, the gradient in registers, one slab per workgroup: 72 us per product at 65 536 rows against 162),
 * everything else on the general gradient kernel; the two produce the same bits (environment OSA_FVP_FAST=0 forces the
 * general kernel). */
int osa_actor_fvp_raw(int obs_dim, int act_dim, int hidden, float* params, float* grads,
                      const float* obs, int ld_obs, long M, const float* vec, int max_blocks,
                      float* ws, float* step_stats, void* stream);

/* out = raw + damping * v  (natural_pg.py:119) plus the analytic log_std block of the Fisher matrix:
 * out[ls_off + d] += ls_coef * v[ls_off + d], ls_coef = 2 / act_dim.  (With world_size > 1 the caller
 * averages `raw` across ranks first: distributed.avg_tensor, natural_pg.py:112.) */
int osa_fvp_finish(int n, const float* raw, const float* v, float damping, int ls_off, int ls_n,
                   float ls_coef, float* out, void* stream);

/* conjugate_gradients (omnisafe/utils/math.py:116-132) split at the Fisher-vector product:
 *   osa_cg_init: x = 0, r = b (F(0) = 0), p = r, scal[0] = r.r, scal[1] = 0 (not converged)
 *   osa_cg_step(z = F p): alpha = r.r/(p.z + eps); x += alpha p; r -= alpha z; if sqrt(r.r) <
 *     residual_tol set scal[1] = 1 (the reference's `break`; later calls are no-ops) else
 *     p = r + (r.r_new/(r.r + eps)) p.   scal: 4 floats on the device. */
int osa_cg_init(int n, const float* b, float* x, float* r, float* p, float* scal, void* stream);
int osa_cg_step(int n, const float* z, float* x, float* r, float* p, float* scal, float residual_tol,
                float eps, void* stream);

/* out = a*x + b*y (y may be NULL) and *out = x.y: step-direction algebra of trpo.py:202-222 and
 * cpo.py:284-337, parameter candidates theta_old + frac * step of the line searches. */
int osa_vec_lincomb(int n, float a, const float* x, float b, const float* y, float* out, void* stream);
int osa_vec_dot(int n, const float* x, const float* y, float* out, void* stream);

/* Full-batch evaluation of a candidate actor for the line searches (TRPO._search_step_size
 * trpo.py:102-138, CPO._cpo_search_step cpo.py:114-171): out4 = { loss_pi = -mean(ratio * adv) with
 * adv = (adv_r - lambda adv_c)/(1 + lambda), loss_cost = mean(ratio * adv_c), KL(old || new).mean(),
 * mean ratio }.  ws: at least 4096 doubles. */
int osa_actor_eval(int obs_dim, int act_dim, int hidden, const float* actor_params, const float* obs,
                   int ld_obs, long M, const float* act, int ld_act, const float* logp,
                   const float* adv_r, const float* adv_c, const float* lagrange,
                   const float* old_mean, int ld_old, const float* old_log_std, double* ws,
                   float* out4, void* stream);

/* The minibatch shuffles of an update: perm[row][0 .. M) = a pseudo-random permutation of 0 .. M-1 per row, one row
 * per pass (the DataLoader(shuffle=True) the reference iterates in every pass of _update:
 * algorithms/on_policy/base/policy_gradient.py:357-377, natural_pg.py:196-223 -- torch's RandomSampler = randperm).
 * No sort: a keyed bijection of [0, 2^k) (24 alternating Feistel steps over the two halves of k = ceil(log2 M) bits,
 * cycle-walked into [0, M)), element i of row r depends on (row_seeds[r], i) only.  row_seeds: device int64[rows]
 * (drawn by the host framework's seeded generator); perm: device int64 [rows][M].  M <= 2^40, rows <= 65535. */
int osa_shuffle_rows(const long long* row_seeds, int rows, long M, long long* perm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OMNISAFE_AMD_H */
