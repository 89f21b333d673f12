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
/* 'vtrace' (onpolicy_buffer.py:312-326,338-405) is not implemented: OSA_EUNSUPPORTED */

const char* osa_strerror(int code);
int osa_version(void);            /* ABI version, currently 1 */
const char* osa_build_arch(void); /* "gfx950" */

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

#ifdef __cplusplus
}
#endif
#endif /* OMNISAFE_AMD_H */
