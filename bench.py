#!/usr/bin/env python
"""Headline benchmark: env-steps/sec (rollout + update), PPO-Lag on SafetyPointGoal1 shapes.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python bench.py --gpus N --steps K --warmup W      # launches its own N ranks (one per GPU, RCCL), like the
                                                       # reference's fork(): omnisafe/utils/distributed.py:121-137
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # (same thing, ranks started by torchrun)

A "step" is ONE EPOCH of the hot path on one batch of synthetic input: rollout of T = 16 vector steps
over 4096 device-resident envs per GPU (65 536 env-steps per GPU), dual GAE + standardisation, then the
PPO-Lag update with the reference's YAML defaults (batch_size 64, update_iters 40 -> 40 960 sequential
optimiser steps x 3 networks) with kl_early_stop OFF so that every epoch does the maximum work (the
reference would usually stop earlier).  This is BASELINE.json configs[1] ("PPOLag on
SafetyPointGoal1-v0, 1xMI355X, 4096 vectorized envs, steps_per_epoch=65536"); weak scaling: each rank
owns 4096 envs, steps_per_epoch = 65 536 x world_size; under data parallelism the rollouts are all-gathered
once per epoch and every rank runs the whole global optimiser chain (no per-step collective, profiles/HISTORY.md §5).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     dominant kernel (osa_ppo_pass_kernel: one persistent launch = one whole pass of 1024 dependent
               64-row optimiser steps of all three networks), algorithmic FLOPs / mean launch time measured
               with HIP events in the timed region
  cpu_baseline the UNMODIFIED REFERENCE (kind "reference": omnisafe.Agent('PPOLag', device=cpu).learn() from the staged
               archive oracle/_ref/omnisafe_ref.zip) timed on this box's host cores on a bounded sample of the same
               workload (--ref-sample-iters of the 40 passes executed, the passes scaled; `full_epoch_value` = the
               committed all-40-passes measurement); cpu_baseline_port = the oracle port (oracle/np_oracle.py) beside it
and a "throughput_variant" object: the same workload shape with the large-batch setting the reference
itself uses for GPU-resident envs (PPOLag.yaml ShadowHand* blocks: batch_size 8192 -> here 16 384,
update_iters 8), which shows what the kernels do when the optimiser chain is not 64 rows wide.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC for RCCL; must precede HIP initialisation

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

OBS_DIM, ACT_DIM, HIDDEN = 60, 2, 64
W_PI = OBS_DIM * HIDDEN + HIDDEN * HIDDEN + HIDDEN * ACT_DIM  # 8064 weights (no biases)
W_V = OBS_DIM * HIDDEN + HIDDEN * HIDDEN + HIDDEN * 1         # 8000
FLOPS_PER_SAMPLE_STEP = 6 * (W_PI + 2 * W_V)                  # fwd 2W + bwd 4W, three nets
PEAK_F32_MFMA_TFLOPS = 157.3                                   # MI355X_MICROARCH.md (f32-input MFMA)
PEAK_HBM_GBPS = 8000.0                                         # MI355X_MICROARCH.md (HBM3E)
# Serial matrix-pipe floor of ONE 64-row optimiser step of one network (osa_ppo_pass_kernel, 60/2): every wave
# issues 320 dependent-chain v_mfma_f32_16x16x4_f32 per step (layer 1: 64, layer 2: 64, dz1: 64, dW2: 64,
# dW1: 64; the 1-2-output layer runs on the VALU), 32 issue cycles each, at the 2.4 GHz engine clock; the three
# networks run concurrently on three CUs.  The B = 64 chain cannot go below this however the rest is scheduled.
MFMA_PER_STEP, MFMA_CYCLES, ENGINE_GHZ = 320, 32, 2.4
W_PI_RUN, W_V_RUN = W_PI, W_V  # (set from --hidden-sizes in main)
GENERAL_WEIGHTS = (0, 0)       # (all weights of the three networks, those above layer 0): set with --hidden-sizes
GENERAL_IS_1024 = False


def weights_of(hidden):
    """(W_pi, W_V): weight counts without biases of the actor / one critic for the given hidden sizes."""
    sizes = [OBS_DIM] + list(hidden)
    body = sum(a * b for a, b in zip(sizes[:-1], sizes[1:]))
    return body + sizes[-1] * ACT_DIM, body + sizes[-1]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--warmup', type=int, default=2)  # epoch 1 eager, epoch 2 captures the rollout hipGraph
    ap.add_argument('--envs', type=int, default=4096)
    ap.add_argument('--steps-per-env', type=int, default=16)
    ap.add_argument('--batch-size', type=int, default=64)
    ap.add_argument('--update-iters', type=int, default=40)
    ap.add_argument('--kl-early-stop', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-variant', action='store_true')
    ap.add_argument('--algo', default='PPOLag')
    # hidden layers of actor and critics (YAML default 64 64; e.g. 1024 1024 = the large-network row of the
    # reference's published timing table, docs/source/start/efficiency.rst:15-23: the layer-wise GEMM path)
    ap.add_argument('--hidden-sizes', type=int, nargs='*', default=[64, 64])
    ap.add_argument('--variant-batch', type=int, default=16384)
    # passes of the reference's update actually executed by the cpu_baseline leg (of --update-iters): 4 passes
    # + the rollout are ~13 s on the GPU box's host (2.7 s rollout + 2.55 s per pass at 16 threads)
    ap.add_argument('--ref-sample-iters', type=int, default=4)
    # data-parallel update mode of the headline line under --gpus N > 1 (omnisafe_amd/update.py): 'replicated' (default:
    # rollouts all-gathered once per epoch, every GPU runs the whole global optimiser chain, no per-step collective),
    # 'replicated-steps', 'allreduce' (per optimiser step: gradient kernel -> ONE flat RCCL all-reduce -> Adam: the
    # reference's structure, policy_gradient.py:437-443).  With N > 1 the line also carries an `allreduce_mode` object
    # (the same workload in the per-step all-reduce mode, --allreduce-steps epochs) unless --no-allreduce-leg, and the
    # large-batch `throughput_variant` (one flat RCCL all-reduce per step inside the captured update graph).
    # 'p2p' (round 6): every rank runs the single-GPU persistent pass on its own rows; clipped gradients exchanged by
    # one-shot peer writes into hipIpc-mapped buffers (osa_ppo_p2p_pass), no collective on the step path
    ap.add_argument('--dp-mode', default=None, choices=['replicated', 'replicated-steps', 'allreduce', 'p2p'])
    ap.add_argument('--no-p2p-leg', action='store_true')
    ap.add_argument('--no-early-stop-leg', action='store_true')
    ap.add_argument('--allreduce-steps', type=int, default=2)
    # passes per epoch of the all-reduce leg: a collective per 64-row optimiser step makes 40 passes (40 960 steps) a
    # minutes-long epoch on slow transports (the one-device gloo hook); its per-step time does not depend on the count
    ap.add_argument('--allreduce-update-iters', type=int, default=4)
    ap.add_argument('--no-allreduce-leg', action='store_true')
    # the N = 1 value of the same command (env-steps/s): adds efficiency_vs_n1 = value / (N x n1) to the line
    ap.add_argument('--n1-value', type=float, default=None)
    ap.add_argument('--n1-variant-value', type=float, default=None)
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` (N > 1) WITHOUT torchrun's environment: start the N ranks ourselves, as the reference's
    `fork` does (omnisafe/utils/distributed.py:121-137 re-executes the script under `torchrun --nproc_per_node N`).
    One process per GPU over `nccl` (= RCCL); when the box has fewer than N devices the launch is refused unless the
    test hook OSA_SINGLE_DEVICE_RANKS=1 is set (all ranks on cuda:0 over gloo: a code-path check, not a measurement --
    the printed line then says rccl_ranks 0)."""
    import socket
    import subprocess

    env = dict(os.environ)
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and not env.get('OSA_SINGLE_DEVICE_RANKS'):
        print(f'bench.py: --gpus {args.gpus} but this box exposes {ndev} GPU(s) (set OSA_SINGLE_DEVICE_RANKS=1 to run '
              f'{args.gpus} ranks on one device over gloo as a code-path check)', file=sys.stderr)
        return 2
    if env.get('OSA_SINGLE_DEVICE_RANKS'):
        env.setdefault('OSA_DIST_BACKEND', 'gloo')  # RCCL refuses two ranks on one device
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env.setdefault('OMP_NUM_THREADS', '1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def make_algo(args, world, batch_size, update_iters, epochs, log_dir):
    import omnisafe_amd

    spe = args.envs * args.steps_per_env * world
    cfg = {
        'seed': 0,
        'train_cfgs': {'device': 'cuda:0' if os.environ.get('OSA_SINGLE_DEVICE_RANKS') else
                       f'cuda:{int(os.environ.get("LOCAL_RANK", 0))}', 'vector_env_nums': args.envs,
                       'total_steps': spe * epochs},
        'algo_cfgs': {'steps_per_epoch': spe, 'batch_size': batch_size, 'update_iters': update_iters,
                      'kl_early_stop': bool(args.kl_early_stop)},
        'logger_cfgs': {'log_dir': log_dir, 'save_model_freq': 10 ** 9, 'verbose': False},
        # horizon <= T so that episodes finish inside an epoch (the reference resets envs every epoch,
        # onpolicy_adapter.py:80; with longer episodes EpCost is empty and ppo_lag.py:74 asserts)
        'env_cfgs': {'horizon': args.steps_per_env, 'cost_p': 0.05},
    }
    if list(args.hidden_sizes) != [64, 64]:
        cfg['model_cfgs'] = {'actor': {'hidden_sizes': list(args.hidden_sizes)},
                             'critic': {'hidden_sizes': list(args.hidden_sizes)}}
    return omnisafe_amd.Agent(args.algo, 'SynthPointGoal1-v0', custom_cfgs=cfg).agent


def run_epochs(algo, n, sync):
    """The body of PolicyGradient.learn()'s epoch loop (policy_gradient.py:252-292)."""
    for _ in range(n):
        algo._env.rollout(steps_per_epoch=algo._steps_per_epoch, agent=algo._actor_critic,
                          buffer=algo._buf, logger=algo._logger)
        algo._update()
        if algo._cfgs.model_cfgs.actor.lr is not None:
            algo._actor_critic.actor_scheduler.step()
        algo._logger.dump_tabular()
    sync()


def timed(algo, steps, warmup, world, dev):
    import torch.distributed as dist

    def sync():
        torch.cuda.synchronize(dev)

    run_epochs(algo, warmup, sync)
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    run_epochs(algo, steps, sync)
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    return dt


def pmc_traffic(kernel_prefixes, tag):
    """HBM-side bytes per launch of the dominant kernel(s) from the committed rocprofv3 PMC passes of THIS command
    (profiles/r<N>_pmc_traffic_<tag>.json: 2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes, gfx950 correction --
    tools/gpu_round_check.sh + tools/pmc_summary.py); PMC counters cannot be read from inside the process, so this is
    the offline measurement of the same command.  The file records the digest of the kernel sources it was measured on
    (`_abi_digest` = osa_abi_digest()); a file measured on OTHER kernels is stale and yields None (round-3 verdict:
    the traffic of a previous build must not ride along with a new kernel's time).  Several prefixes = a step made of
    several launches: their sum."""
    import glob

    try:
        from omnisafe_amd import build as _b

        digest = _b.source_digest()
    except Exception:  # noqa: BLE001
        return None
    d = None
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r*_pmc_traffic_{tag}.json')), reverse=True):
        cand = json.load(open(path))
        if cand.get('_abi_digest') == digest:  # (the newest round's file measured on exactly these kernel sources)
            d = cand
            break
    if d is None:
        return None
    total = 0
    for pre in kernel_prefixes:
        hit = [v['traffic_bytes_per_launch'] for k, v in d.items() if k.startswith(pre)]
        if len(hit) != 1:
            return None
        total += hit[0]
    return total


def roofline_from_events(events, batch_size):
    """Dominant-kernel roofline from HIP events recorded around every update launch inside the timed
    region (torch.cuda.Event on the launch stream).  B <= 64: osa_ppo_pass_kernel, one launch = one
    whole pass of ceil(M/B) dependent optimiser steps of the three networks (3 workgroups).
    Larger B: osa_mb_grad_kernel (+ the two small reduce/finalize launches it is bracketed with)."""
    name = events[0][0]
    rows = sum(e[1] for e in events)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    flops_per_sample = 6 * (W_PI_RUN + 2 * W_V_RUN)
    if name in ('osa_ppo_dp_step', 'osa_ppo_dp_pass'):
        rows //= world  # algorithmic work of a rank = its own M rows (it executes world x that, redundantly)
    ms = sum(e[2][0].elapsed_time(e[2][1]) for e in events)
    flops = flops_per_sample * rows
    achieved = flops / (ms * 1e-3) / 1e12
    us = ms * 1e3 / len(events)
    if name == 'osa_mb_grad_kernel':
        traffic = pmc_traffic(['osa_ppo_part_kernel', 'osa_slab_reduce_finalize_kernel'], 'variant')
    elif name == 'osa_ppo_pass_kernel' and batch_size <= 64:
        traffic = pmc_traffic(['osa_ppo_pass_kernel'], 'bench')
    else:
        traffic = None
    out = {'bound': 'mfma', 'achieved': round(achieved, 4), 'peak': PEAK_F32_MFMA_TFLOPS,
           'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 5),
           'traffic': traffic,
           'kernel': name, 'launches_timed': len(events), 'us_per_launch': round(us, 2),
           'flops_per_launch': flops // len(events), 'rows_per_launch': rows // len(events)}
    if name == 'osa_ppo_dp_step':
        out['kernel'] = 'osa_ppo_pass_kernel (data-parallel gradient mode) + osa_dp_apply_kernel'
        out['note'] = ('replicated-data DP: each rank computes all ranks\' minibatches of every optimiser step '
                       '(W x 3 workgroups) on the all-gathered rollout, hipGraph of one pass replayed; rows = W x M')
    elif name == 'osa_ppo_dp_pass':
        steps = rows // len(events) // batch_size
        out['kernel'] = 'osa_ppo_pass_kernel<..., COOP> (cooperative data-parallel pass)'
        out['us_per_optimiser_step'] = round(us / steps, 3)
        out['note'] = (f'replicated-data DP, one persistent launch per pass: 3 x {world} resident workgroups, each '
                       f'computing one rank\'s {batch_size}-row minibatch of every one of the {steps} global optimiser '
                       'steps on the all-gathered rollout, exchanging clipped gradients through an agent-scope '
                       'arrival counter; flops counted for this rank\'s own rows only')
    elif name == 'osa_ppo_pass_kernel':
        steps = rows // len(events) // batch_size
        out['us_per_optimiser_step'] = round(us / steps, 3)
        if batch_size <= 64:
            floor = MFMA_PER_STEP * MFMA_CYCLES / (ENGINE_GHZ * 1e3)
            out['chain_floor_us'] = round(floor, 3)  # serial MFMA issue cycles of one step (see the constants)
            out['chain_frac'] = round(floor / (us / steps), 4)  # the meaningful denominator of a B = 64 chain
            # (why the step sits at ~2 x the floor: float32 MFMA and VALU work do not overlap on gfx950 -- raw probe
            # output and the per-phase clocks of the current step)
            out['chain_floor_evidence'] = ['profiles/r6_mfma_overlap_probe_raw.txt', 'profiles/r6_pass_phase_clocks.txt']
        out['note'] = (f'persistent pass: {steps} dependent {batch_size}-row optimiser steps per launch on 3 '
                       'workgroups (one per network) of a 256-CU chip -- latency-bound by the reference\'s '
                       'batch_size=64 chain, not by MFMA throughput; see throughput_variant')
    elif name == 'gm_gemm_kernel' and batch_size > 64:
        steps = max(1, rows // len(events) // batch_size)
        out['us_per_optimiser_step'] = round(us / steps, 3)
        out['kernel'] = 'gm_gemm_kernel (csrc/general_mlp.hip: forward / backward-data / backward-weight GEMMs) + loss / reduce / Adam'
        out['note'] = ('general networks, layer-wise: ~13 launches per optimiser step on one float32-MFMA GEMM kernel '
                       '(v_mfma_f32_32x32x2_f32), three networks per launch; the timed unit is a whole step (or captured pass)')
    elif name == 'gm_gemm_kernel' and batch_size <= 64:
        # general networks at the YAML batch: the skinny kernels (csrc/skinny_mlp.h) -- a 64-row step is bandwidth work.
        # Algorithmic HBM bytes per optimiser step: every weight read by the forward pass, every weight above layer 0 by
        # the backward-data pass, weights and both Adam moments read and written once: 4 (7 W + W_above_0) bytes.
        steps = max(1, rows // len(events) // batch_size)
        w_all, w_up = GENERAL_WEIGHTS
        alg_bytes = 4 * (7 * w_all + w_up)
        us_step = us / steps
        gbs = alg_bytes / (us_step * 1e-6) / 1e9
        traffic = pmc_traffic(['gs_fwd_kernel<true>', 'gs_fwd_kernel<false>', 'gs_top_kernel', 'gs_bwd_kernel',
                               'gs_wgrad_kernel<1>'], 'general_1024_B64') if GENERAL_IS_1024 else None
        out = {'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s',
               'frac': round(gbs / PEAK_HBM_GBPS, 5), 'traffic': traffic,
               'kernel': 'gs_fwd_kernel x (L - 1) + gs_top_kernel + gs_bwd_kernel x (L - 2) + gs_wgrad_kernel<1> '
                         '(csrc/skinny_mlp.h; L linear layers, three networks per launch)',
               'launches_timed': len(events), 'us_per_optimiser_step': round(us_step, 3),
               'algorithmic_bytes_per_step': alg_bytes, 'traffic_per': 'optimiser step (sum of the five launches)',
               'flops_per_step': flops // len(events) // steps,
               'note': ('general networks, minibatches of <= 64 rows: 2 L - 1 launches per optimiser step (clip norm from '
                        'Gram matrices inside the forward / top / backward launches, top layer fused into the launch '
                        'below); bound by streaming the weights and the Adam state, not by the matrix pipe')}
    else:
        steps = max(1, rows // len(events) // batch_size)  # (a captured pass of several steps is one timed event)
        out['us_per_optimiser_step'] = round(us / steps, 3)
        out['kernel'] = 'osa_ppo_part_kernel (balanced partial gradients) + osa_slab_reduce_finalize_kernel'
        out['steps_per_timed_launch'] = steps  # the timed unit is a captured graph of `steps` optimiser steps
        out['traffic_per'] = 'optimiser step (one launch of each of the two kernels); algorithmic: batch_size x 268 B'
        out['note'] = ('large-batch step: the 64-row chunk-tasks of the three networks shared evenly by one workgroup '
                       'per compute unit (weights in LDS, partial gradient in registers, one slab per segment); slab '
                       'reduce + clip + Adam in a second launch')
    return out


def whole_path(args, value, world, update_iters, kl_passes):
    """SURVEY.md 8(d) whole-path figures of the headline metric: algorithmic bytes and FLOPs per env-step
    (rollout + GAE + standardise side: 4 D_o + 4 D_a + 72 B; update: K x (4 (D_o + D_a + 5) B gathered +
    6 (W_pi + 2 W_V) FLOP) per env-step, + one full-batch KL pass (4 D_o B, 2 W_pi FLOP) per pass that runs one) and
    what the measured env-steps/s turn them into, against the vendor peaks.  Per GPU."""
    bytes_step = (4 * OBS_DIM + 4 * ACT_DIM + 72) + update_iters * 4 * (OBS_DIM + ACT_DIM + 5) + kl_passes * 4 * OBS_DIM
    flops_step = update_iters * 6 * (W_PI_RUN + 2 * W_V_RUN) + kl_passes * 2 * W_PI_RUN
    per_gpu = value / world
    gbps, tflops = bytes_step * per_gpu / 1e9, flops_step * per_gpu / 1e12
    return {'bytes_per_env_step': bytes_step, 'flops_per_env_step': flops_step,
            'achieved_GBps': round(gbps, 3), 'frac_hbm_peak': round(gbps / PEAK_HBM_GBPS, 6),
            'achieved_TFLOPs': round(tflops, 4), 'frac_f32_mfma_peak': round(tflops / PEAK_F32_MFMA_TFLOPS, 5),
            'kl_passes_per_epoch': kl_passes,
            'note': 'algorithmic (minimum) bytes / FLOPs of the whole epoch per env-step x measured env-steps/s per GPU'}


def cpu_baseline(args):
    """Oracle (CPU restatement of the reference, bit-exact to it) on a bounded sample of the same
    epoch: the full rollout (65 536 transitions: per-env loop, normaliser, policy step, GAE, get)
    plus `n_mb` of the 40 960 minibatch optimiser steps and one full-batch KL pass; the update time is
    scaled to 40 passes."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import np_oracle as O

    # threads: the reference's own default torch_threads (PPOLag.yaml: 16), capped by the host; the
    # 64-wide MLPs do not scale past a few cores and over-subscription on many-core hosts is
    # pathological, so "all cores" is not used.
    cores = max(1, min(os.cpu_count() or 1, 16))
    torch.set_num_threads(cores)
    N, T = args.envs, args.steps_per_env
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    ac = O.ActorCritic(OBS_DIM, ACT_DIM)
    norm = O.Normalizer((OBS_DIM,), clip=5)
    trunc = np.zeros((T, N), bool)
    trunc[T - 1] = True
    trace = {'reset_obs': rng.standard_normal((N, OBS_DIM)).astype(np.float32),
             'obs': rng.standard_normal((T, N, OBS_DIM)).astype(np.float32),
             'reward': rng.standard_normal((T, N)).astype(np.float32),
             'cost': (rng.random((T, N)) < 0.05).astype(np.float32),
             'terminated': np.zeros((T, N), bool), 'truncated': trunc,
             'final_obs': rng.standard_normal((T, N, OBS_DIM)).astype(np.float32),
             'eps': rng.standard_normal((T, N, ACT_DIM)).astype(np.float32)}
    t0 = time.perf_counter()
    buf, gae, _aux = O.rollout_on_trace(ac, norm, trace)
    a_r, a_c, _ = O.buffer_get(gae['adv_r'], gae['adv_c'])
    t_roll = time.perf_counter() - t0
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    data = {'obs': tt(O.env_major(buf['obs'])), 'act': tt(O.env_major(buf['act'])),
            'logp': tt(O.env_major(buf['logp'])), 'target_value_r': tt(O.env_major(gae['tgt_r'])),
            'target_value_c': tt(O.env_major(gae['tgt_c'])), 'adv_r': tt(a_r), 'adv_c': tt(a_c)}
    M, B = N * T, args.batch_size
    perm = torch.randperm(M)
    budget_s, n_mb = 10.0, 0
    t0 = time.perf_counter()
    for s in range(0, M - B + 1, B):  # time-bounded sample of the minibatch chain
        idx = perm[s:s + B]
        O.critic_step(ac.reward_critic, ac.reward_critic_optimizer, data['obs'][idx], data['target_value_r'][idx])
        O.critic_step(ac.cost_critic, ac.cost_critic_optimizer, data['obs'][idx], data['target_value_c'][idx])
        O.actor_step(ac.actor, ac.actor_optimizer, data['obs'][idx], data['act'][idx], data['logp'][idx],
                     data['adv_r'][idx], data['adv_c'][idx], 0.001)
        n_mb += 1
        if time.perf_counter() - t0 > budget_s:
            break
    per_mb = (time.perf_counter() - t0) / n_mb
    with torch.no_grad():
        old = ac.actor.dist(data['obs'])
        om, osd = old.mean.clone(), old.stddev.clone()
    t0 = time.perf_counter()
    O.kl_old_new(ac.actor, data['obs'], om, osd)
    t_kl = time.perf_counter() - t0
    nmb_full = (M + B - 1) // B
    t_epoch = t_roll + args.update_iters * (nmb_full * per_mb + t_kl)
    return {'value': round(M / t_epoch, 1), 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port',
            'sample': (f'full rollout+GAE+get of {M} transitions ({t_roll:.2f} s) + {n_mb} of '
                       f'{args.update_iters * nmb_full} minibatch optimiser steps ({per_mb * 1e3:.3f} ms '
                       f'each) + 1 KL pass ({t_kl * 1e3:.1f} ms); update extrapolated to '
                       f'{args.update_iters} passes'),
            'rollout_s': round(t_roll, 3), 'ms_per_minibatch_step': round(per_mb * 1e3, 4)}


def cpu_baseline_reference(args):
    """The UNMODIFIED reference on this box's host cores (kind "reference"): oracle/ref_cpu_baseline.py runs
    `omnisafe.Agent('PPOLag', ..., device=cpu).learn()` for one epoch of this benchmark's shape with
    update_iters lowered to --ref-sample-iters (the bounded sample) in its own process and reads Time/Rollout / Time/Update /
    Time/FPS from the reference's progress.csv.  The package comes from /root/reference or from the archive
    staged by `__graft_entry__.build()` (oracle/_ref/omnisafe_ref.zip); None if neither exists."""
    import subprocess

    threads = max(1, min(os.cpu_count() or 1, 16))  # the reference's own default torch_threads (PPOLag.yaml)
    cmd = [sys.executable, os.path.join(ROOT, 'oracle', 'ref_cpu_baseline.py'), '--envs', str(args.envs),
           '--steps-per-env', str(args.steps_per_env), '--batch-size', str(args.batch_size),
           '--update-iters', str(args.update_iters), '--sample-iters', str(args.ref_sample_iters), '--threads', str(threads),
           '--algo', args.algo]
    try:
        env = dict(os.environ, OMP_NUM_THREADS=str(threads))
        for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
            env.pop(k, None)
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1]
        res = json.loads(line)
        return None if 'error' in res else res
    except Exception as exc:  # noqa: BLE001 - the baseline is a reported extra, never fatal
        return {'error': f'{type(exc).__name__}: {exc}'}


def full_epoch_reference(args):
    """The unmodified reference's env-steps/s with ALL update passes executed, measured on a GPU box's host by this
    script's own `--ref-sample-iters <update_iters>` leg and committed (profiles/r*_reference_full_epoch_config2.json);
    only for the default workload the file was measured on."""
    import glob

    if (args.algo, args.envs, args.steps_per_env, args.batch_size, args.update_iters, list(args.hidden_sizes)) != (
            'PPOLag', 4096, 16, 64, 40, [64, 64]):
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_reference_full_epoch_config2.json')), reverse=True):
        try:
            d = json.load(open(path))
            if int(d.get('sample_iters', 0)) == args.update_iters:
                return {'value': d['value'], 'source': os.path.relpath(path, ROOT)}
        except Exception:  # noqa: BLE001
            continue
    return None


def kl_pass_us(algo, dev, reps=20):
    """HIP-event time of ONE full-batch KL(old || new) pass (policy_gradient.py:383-390) on this rank's rollout."""
    up = algo._updater
    try:
        obs = algo._last_update_data['obs']  # the rows of the last epoch's update (env-major `buf.get()` output)
        up.snapshot_old_distribution(obs)
        up.kl(obs)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            up.kl(obs)
        e1.record()
        torch.cuda.synchronize(dev)
        return round(e0.elapsed_time(e1) * 1e3 / reps, 2)
    except Exception:  # noqa: BLE001 - a reported extra
        return None


def main():
    args = parse()
    if args.gpus > 1 and 'RANK' not in os.environ and int(os.environ.get('WORLD_SIZE', '1')) == 1:
        sys.exit(self_launch(args))  # the driver's `python bench.py --gpus N`: launch the ranks ourselves
    if args.dp_mode:
        os.environ['OSA_DP_MODE'] = args.dp_mode
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('OSA_SINGLE_DEVICE_RANKS'):  # test hook: several ranks on one GPU (gloo)
        local = 0
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    dev = torch.device(f'cuda:{local}')
    torch.cuda.set_device(dev)
    import tempfile

    global W_PI_RUN, W_V_RUN, GENERAL_WEIGHTS, GENERAL_IS_1024
    W_PI_RUN, W_V_RUN = weights_of(args.hidden_sizes)
    _first = OBS_DIM * args.hidden_sizes[0] if args.hidden_sizes else 0
    GENERAL_WEIGHTS = (W_PI_RUN + 2 * W_V_RUN, W_PI_RUN + 2 * W_V_RUN - 3 * _first)
    GENERAL_IS_1024 = list(args.hidden_sizes) == [1024, 1024] and args.batch_size == 64
    general = list(args.hidden_sizes) != [64, 64]
    hid = 'x'.join(str(h) for h in args.hidden_sizes)
    log_dir = tempfile.mkdtemp(prefix='osa_bench_')
    algo = make_algo(args, world, args.batch_size, args.update_iters, args.steps + args.warmup + 1, log_dir)
    events, fvp_events = [], []
    run_epochs(algo, args.warmup, lambda: torch.cuda.synchronize(dev))
    algo._updater.profile_events = events
    if hasattr(algo, '_solver'):  # trust-region family (--algo CPO / TRPOLag): HIP events around every Fisher-vector product
        algo._solver.profile_events = fvp_events
    dt = timed(algo, args.steps, 0, world, dev)
    algo._updater.profile_events = None
    if hasattr(algo, '_solver'):
        algo._solver.profile_events = None
    per_gpu_steps = args.envs * args.steps_per_env
    value = world * per_gpu_steps * args.steps / dt
    out = {
        'metric': 'env-steps/sec (rollout+update), PPO-Lag SafetyPointGoal1',
        'value': round(value, 1), 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': (f'{args.algo}, SafetyPointGoal1 shapes (obs 60, act 2, {hid} tanh), '
                                f'{args.envs} vectorized envs/GPU, steps_per_epoch={per_gpu_steps}/GPU '
                                f'(T={args.steps_per_env}), batch_size={args.batch_size}, '
                                f'update_iters={args.update_iters}, kl_early_stop='
                                f'{bool(args.kl_early_stop)}; 1 step = 1 epoch (rollout+GAE+update)'),
                   'env_steps_per_step': world * per_gpu_steps, 'parallelism': f'dp{world}',
                   # (explicit keys: the driver's record truncates `workload`)
                   'algo': args.algo, 'obs_dim': OBS_DIM, 'act_dim': ACT_DIM, 'hidden_sizes': list(args.hidden_sizes),
                   'envs_per_gpu': args.envs, 'steps_per_env': args.steps_per_env,
                   'steps_per_epoch_per_gpu': per_gpu_steps, 'batch_size': args.batch_size,
                   'update_iters': args.update_iters, 'kl_early_stop': bool(args.kl_early_stop),
                   'dp_mode': (os.environ.get('OSA_DP_MODE', 'replicated') if world > 1 else None),
                   'update_path': algo._updater.last_path,
                   'rollout_path': ('graph of launches' if getattr(algo._env, 'last_rollout_graphed', False)
                                    else getattr(algo._env, 'last_rollout_path', 'launches'))},
    }
    if args.algo != 'PPOLag':
        out['metric'] = f'env-steps/sec (rollout+update), {args.algo} SafetyPointGoal1 shapes'
    import torch.distributed as tdist

    backend = tdist.get_backend() if tdist.is_initialized() else None
    # ranks that talk over RCCL (`nccl` backend, one device each); 0 = single process, or the one-device gloo test hook
    out['dist_backend'] = backend
    out['rccl_ranks'] = tdist.get_world_size() if backend == 'nccl' else 0
    out['devices_visible'] = torch.cuda.device_count()
    if args.n1_value:
        out['efficiency_vs_n1'] = round(value / (world * args.n1_value), 4)
    # (the trust-region family updates critics only in the minibatch loop: its events carry 2 of the 3 networks)
    out['roofline'] = roofline_from_events(events, int(algo._updater.batch_size))
    if hasattr(algo, '_solver') and not algo._updater.update_actor:
        r = out['roofline']
        scale = (4 * W_V_RUN) / (2 * (W_PI_RUN + 2 * W_V_RUN))  # critics only: 6 x 2 W_V of the 6 (W_pi + 2 W_V)
        r['achieved'] = round(r['achieved'] * scale, 4)
        r['frac'] = round(r['frac'] * scale, 5)
        r['flops_per_launch'] = int(r['flops_per_launch'] * scale)
        r['note'] = 'critic passes of the trust-region family (actor excluded from the FLOP count); ' + r.get('note', '')
    if fvp_events:
        # Fisher-vector product (natural_pg.py:91-119 as JVP -> VJP, profiles/HISTORY.md §3.1): forward 2 W, tangent forward 2 x 2 W
        # (two products per layer), backward 4 W = 8 W_pi FLOP... per row the count the round-3 verdict prescribes
        ms = sum(e[2][0].elapsed_time(e[2][1]) for e in fvp_events)
        rows = sum(e[1] for e in fvp_events)
        fl = 8 * W_PI_RUN * rows
        ach = fl / (ms * 1e-3) / 1e12
        out['roofline_fvp'] = {
            'bound': 'mfma', 'achieved': round(ach, 4), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(ach / PEAK_F32_MFMA_TFLOPS, 5),
            # (HBM-side bytes of one product from the committed PMC passes of `bench.py --algo <algo>`, digest-checked;
            # algorithmic: rows x 4 D_o bytes of observations + the gradient slabs)
            'traffic': (pmc_traffic(['osa_fvp_kernel', 'osa_fvp_reduce_kernel'], args.algo)
                        if fvp_events[0][0].startswith('osa_fvp_kernel') else None),
            'kernel': fvp_events[0][0],
            'launches_timed': len(fvp_events), 'us_per_launch': round(ms * 1e3 / len(fvp_events), 2),
            'rows_per_launch': rows // len(fvp_events), 'flops_per_launch': fl // len(fvp_events),
            'fvp_per_epoch': len(fvp_events) // max(args.steps, 1),
            'note': 'one launch = one Fisher-vector product over the whole rollout (8 W_pi FLOP per row); per epoch: '
                    'cg_iters + 2 products (TRPO-Lag) or 2 x cg_iters + 3 (CPO: a second CG for the cost gradient), '
                    'then the line-search evaluations (forward only)'}
    # (without early stop only the last pass's KL is evaluated; with it, one per pass: update.py)
    out['whole_path'] = whole_path(args, value, world, args.update_iters,
                                   args.update_iters if args.kl_early_stop else 1)
    if not args.kl_early_stop and args.algo == 'PPOLag':
        # Work NOT done in the timed region against the reference's loop: the reference evaluates the full-batch KL after
        # EVERY pass (policy_gradient.py:383-390); with kl_early_stop off only the last value is observable (logged
        # Train/KL), so the earlier update_iters - 1 passes are not computed.  Their cost, measured:
        us = kl_pass_us(algo, dev)
        out['config']['kl_passes_skipped'] = args.update_iters - 1
        out['config']['kl_pass_us'] = us
        if us is not None:
            out['config']['kl_passes_skipped_frac_of_epoch'] = round((args.update_iters - 1) * us * 1e-3 /
                                                                      (dt / args.steps * 1e3), 5)
    headline_mode = os.environ.get('OSA_DP_MODE', 'replicated')
    if world > 1 and not args.no_allreduce_leg and headline_mode != 'allreduce' and args.algo == 'PPOLag':
        try:
            # the same workload in the reference's own structure: per optimiser step gradient kernel -> ONE flat RCCL
            # all-reduce of the three networks' clipped gradients -> Adam (policy_gradient.py:437-443 with 1 message for 19);
            # every rank runs only its own 64-row minibatch.  A few epochs: a step waits for the collective's latency.
            del algo
            torch.cuda.empty_cache()
            os.environ['OSA_DP_MODE'] = 'allreduce'
            a_iters = min(args.update_iters, args.allreduce_update_iters)
            a_algo = make_algo(args, world, args.batch_size, a_iters, args.allreduce_steps + 3, log_dir)
            run_epochs(a_algo, 2, lambda: torch.cuda.synchronize(dev))
            a_dt = timed(a_algo, args.allreduce_steps, 0, world, dev)  # (no per-step events: 40 960 steps per epoch)
            a_val = world * per_gpu_steps * args.allreduce_steps / a_dt
            n_steps = a_iters * ((per_gpu_steps + args.batch_size - 1) // args.batch_size)
            out['allreduce_mode'] = {
                'workload': (f'same shapes, update_iters={a_iters} (NOT the headline\'s {args.update_iters}: bounded run), '
                             'dp_mode=allreduce (gradient kernel -> flat all-reduce -> Adam per optimiser step)'),
                'update_iters': a_iters,
                'value': round(a_val, 1), 'unit': 'env-steps/s', 'n_gpus': world,
                'ms_per_step': round(a_dt / args.allreduce_steps * 1e3, 3), 'steps': args.allreduce_steps, 'warmup': 2,
                'update_path': a_algo._updater.last_path,
                'us_per_optimiser_step_incl_collective': round(a_dt / args.allreduce_steps * 1e6 / n_steps, 3),
                'message_bytes': (4 * int(a_algo._updater.lib.osa_ppo_dp_ws_floats(OBS_DIM, ACT_DIM, HIDDEN, 1))
                                  if not general else None)}
            if args.n1_value and a_iters == args.update_iters:
                # (a ratio of throughputs is a scaling efficiency only at the SAME update_iters as the N = 1 headline)
                out['allreduce_mode']['efficiency_vs_n1'] = round(a_val / (world * args.n1_value), 4)
            os.environ['OSA_DP_MODE'] = headline_mode
            algo = a_algo
        except Exception as exc:  # noqa: BLE001 -- a secondary leg must never cost the headline its JSON line
            out['allreduce_mode'] = {'error': repr(exc)[:400]}
            os.environ['OSA_DP_MODE'] = headline_mode
            algo = None
    if world > 1 and not args.no_p2p_leg and headline_mode != 'p2p' and args.algo == 'PPOLag' and not general:
        try:
            # the same workload with the one-shot peer exchange (round 6): each rank runs the single-GPU persistent pass on
            # its own rows, the clipped gradients travel by peer writes into hipIpc-mapped buffers -- no collective and no
            # W-fold recomputation on the step path; FULL update_iters (a pass is one launch per rank)
            del algo
            torch.cuda.empty_cache()
            os.environ['OSA_DP_MODE'] = 'p2p'
            # (a peer that never arrives costs every step this long before the sticky time-out word ends the waiting:
            # seconds, not the library's default 20 s per step, in a leg that has never seen two real devices)
            os.environ.setdefault('OSA_P2P_TIMEOUT_S', '3')
            shared = bool(os.environ.get('OSA_SINGLE_DEVICE_RANKS'))
            # ranks that SHARE a device (the 1-GPU code-path check) are time-sliced, not concurrent: a rank's persistent
            # pass spins through its slice until the peer's gets one -- milliseconds per optimiser step, nothing to do
            # with the exchange (tools/p2p_timing.py starts the ranks behind a barrier and takes the best pass).  One pass
            # and one epoch there: the launch, the mapping and the bits are what the check is for
            p_iters, p_steps = (1, 1) if shared else (args.update_iters, args.steps)
            p_algo = make_algo(args, world, args.batch_size, p_iters, p_steps + 3, log_dir)
            run_epochs(p_algo, 1 if shared else 2, lambda: torch.cuda.synchronize(dev))
            p_dt = timed(p_algo, p_steps, 0, world, dev)
            p_val = world * per_gpu_steps * p_steps / p_dt
            n_steps = p_iters * ((per_gpu_steps + args.batch_size - 1) // args.batch_size)
            out['p2p_mode'] = {
                'workload': f'same shapes, update_iters={p_iters}, dp_mode=p2p (osa_ppo_p2p_pass)',
                'value': round(p_val, 1), 'unit': 'env-steps/s', 'n_gpus': world,
                'ms_per_step': round(p_dt / p_steps * 1e3, 3), 'steps': p_steps, 'warmup': 1 if shared else 2,
                'update_path': p_algo._updater.last_path,
                'us_per_optimiser_step_incl_rollout': round(p_dt / p_steps * 1e6 / n_steps, 3)}
            if shared:
                out['p2p_mode']['note'] = ('ranks share ONE device: their persistent passes are time-sliced, the figure '
                                           'says nothing about the exchange (profiles/r6_p2p_timing.json does)')
            if args.n1_value and not shared:
                out['p2p_mode']['efficiency_vs_n1'] = round(p_val / (world * args.n1_value), 4)
            os.environ['OSA_DP_MODE'] = headline_mode
            algo = p_algo
        except Exception as exc:  # noqa: BLE001 -- a secondary leg must never cost the headline its JSON line
            out['p2p_mode'] = {'error': repr(exc)[:400]}
            os.environ['OSA_DP_MODE'] = headline_mode
            algo = None
    if world == 1 and not args.no_early_stop_leg and args.algo == 'PPOLag' and not args.kl_early_stop and not general:
        try:
            # the YAML default `kl_early_stop: true` (PPOLag.yaml; SURVEY 8d's parity variant): one full-batch KL pass AND
            # one host synchronisation per pass.  (a) target_kl at the YAML's 0.02: the update stops where the reference's
            # would (passes_executed < update_iters: less work per epoch, not comparable with the headline); (b) a target
            # nothing reaches: all passes, i.e. the headline's work + update_iters - 1 KL passes + update_iters host syncs
            del algo
            torch.cuda.empty_cache()
            es = {}
            for tag, tkl in (('yaml_target_kl_0.02', None), ('never_stops', 1e30)):
                args.kl_early_stop = True
                e_algo = make_algo(args, world, args.batch_size, args.update_iters, 8, log_dir)
                args.kl_early_stop = False
                if tkl is not None:
                    e_algo._updater.target_kl = tkl
                run_epochs(e_algo, 2, lambda: torch.cuda.synchronize(dev))
                passes = []
                t0 = time.perf_counter()
                for _ in range(3):
                    run_epochs(e_algo, 1, lambda: torch.cuda.synchronize(dev))
                    passes.append(int(getattr(e_algo, '_last_update_steps', 0)) // ((per_gpu_steps + args.batch_size - 1) // args.batch_size))
                e_dt = (time.perf_counter() - t0) / 3
                es[tag] = {'value': round(per_gpu_steps / e_dt, 1), 'ms_per_step': round(e_dt * 1e3, 3),
                           'passes_executed': passes, 'target_kl': float(e_algo._updater.target_kl)}
                del e_algo
                torch.cuda.empty_cache()
            es['note'] = ('kl_early_stop on (PPOLag.yaml default): a KL pass + a host read per pass; `never_stops` = every pass '
                          'executed = the headline workload + the 39 skipped KL passes + 40 syncs')
            out['kl_early_stop_epoch'] = es
            algo = None
        except Exception as exc:  # noqa: BLE001 -- a secondary leg must never cost the headline its JSON line
            out['kl_early_stop_epoch'] = {'error': repr(exc)[:400]}
            os.environ['OSA_DP_MODE'] = headline_mode
            algo = None
    if not args.no_variant and args.algo == 'PPOLag':
        try:
            # the large-batch setting on EVERY rank (round 4): under world_size > 1 the update is the data-parallel
            # large-batch pass -- partial gradients, local clip, ONE flat RCCL all-reduce, Adam per step, the whole pass
            # incl. its collectives one captured hipGraph (`dp-large-batch-graph`)
            algo = None
            torch.cuda.empty_cache()
            vb = args.variant_batch
            v_algo = make_algo(args, world, vb, 8, 14, log_dir)
            v_steps = 10 if not general else 3
            v_events = []
            run_epochs(v_algo, 3, lambda: torch.cuda.synchronize(dev))  # eager epoch, capture epoch, one replay
            v_algo._updater.profile_events = v_events
            v_dt = timed(v_algo, v_steps, 0, world, dev)
            v_algo._updater.profile_events = None
            v_val = world * per_gpu_steps * v_steps / v_dt
            out['throughput_variant'] = {
                'workload': f'same shapes, batch_size={vb}, update_iters=8 (large-batch setting of PPOLag.yaml '
                            'GPU-env blocks)',
                'value': round(v_val, 1), 'unit': 'env-steps/s', 'n_gpus': world,
                'ms_per_step': round(v_dt / v_steps * 1e3, 3), 'steps': v_steps, 'warmup': 3,
                'rollout_graphed': bool(getattr(v_algo._env, 'last_rollout_graphed', False)),
                'rollout_path': getattr(v_algo._env, 'last_rollout_path', 'launches'),
                'update_path': v_algo._updater.last_path,
                'roofline': roofline_from_events(v_events, vb),
                'whole_path': whole_path(args, v_val, world, 8, 1)}
            if args.n1_variant_value:
                out['throughput_variant']['efficiency_vs_n1'] = round(v_val / (world * args.n1_variant_value), 4)
            if world > 1:
                # per-step exchange time = what a step costs beyond the single-GPU step measured on the same kernels
                # (profiles/r4_dp_large_batch_world1_rccl.json: +8.9 us at world 1, no wire time)
                out['throughput_variant']['note'] = (
                    'data-parallel large-batch pass: us_per_optimiser_step includes the flat RCCL all-reduce of '
                    f'{3 * 8448 * 4} bytes and osa_adam_apply of every step; compare with the N = 1 line')
        except Exception as exc:  # noqa: BLE001 -- a secondary leg must never cost the headline its JSON line
            out['throughput_variant'] = {'error': repr(exc)[:400]}
            os.environ['OSA_DP_MODE'] = headline_mode
            algo = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        port = cpu_baseline(args)
        ref = cpu_baseline_reference(args)
        if ref is not None and 'error' not in ref:
            if int(ref.get('sample_iters', 0)) < args.update_iters:
                # the bounded sample extrapolates its passes; the same command with ALL passes executed (105 s on the
                # box's host: `--ref-sample-iters 40`, tools/gpu_round_check.sh stage 6) is the committed measurement
                full = full_epoch_reference(args)
                if full is not None:
                    ref['full_epoch_value'] = full['value']
                    ref['full_epoch_source'] = full['source']
                    ref['gpu_over_full_epoch_reference'] = round(value / full['value'], 1)
            out['cpu_baseline'] = ref           # the unmodified reference on this box's host cores
            out['cpu_baseline_port'] = port     # the oracle port (vectorised numpy rollout: flatters the CPU)
        else:
            out['cpu_baseline'] = port
            out['cpu_baseline_reference_error'] = ref
    else:
        out['cpu_baseline'] = None
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
