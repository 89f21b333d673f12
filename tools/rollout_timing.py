"""Rollout side of an epoch, timed alone: one persistent launch (csrc/rollout_persistent.hip) against the captured
graph of launches and the eager launches, on the benchmark's shapes (4096 envs x 16 steps, obs 60 / act 2).

    python tools/rollout_timing.py [--envs 4096] [--steps-per-env 16] [--epochs 30] > gpurun_out/rollout_timing.json

Per mode: wall time per epoch of adapter.rollout() (host call -> episode metrics on the host, i.e. including the one
synchronisation of the epoch) and the device time between the first and the last kernel of the rollout (events)."""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(mode, args):
    import omnisafe_amd

    os.environ['OSA_ROLLOUT_PERSISTENT'] = '1' if mode.startswith('persistent') else '0'
    os.environ['OSA_ROLLOUT_DEFER_CRITICS'] = '0' if mode.endswith('critics-inside') else '1'
    os.environ['OSA_ROLLOUT_GRAPH'] = '0' if mode == 'launches' else '1'
    N, T = args.envs, args.steps_per_env
    cfg = {'seed': 0, 'train_cfgs': {'device': 'cuda:0', 'vector_env_nums': N, 'total_steps': N * T * 1000},
           'algo_cfgs': {'steps_per_epoch': N * T, 'batch_size': 16384, 'update_iters': 1},
           'logger_cfgs': {'log_dir': tempfile.mkdtemp(), 'save_model_freq': 10 ** 9, 'verbose': False},
           'env_cfgs': {'horizon': T, 'cost_p': 0.05}}
    algo = omnisafe_amd.Agent('PPOLag', 'SynthPointGoal1-v0', custom_cfgs=cfg).agent
    ad = algo._env

    def one():
        ad.rollout(steps_per_epoch=algo._steps_per_epoch, agent=algo._actor_critic, buffer=algo._buf,
                   logger=algo._logger)
        algo._buf.ptr = 0  # (no update between the rollouts: the buffer is simply refilled)

    for _ in range(3):
        one()
    torch.cuda.synchronize()
    clocks = None
    if mode.startswith('persistent'):  # workgroup 0's phase clocks of one epoch (100 MHz ticks)
        from omnisafe_amd import _lib

        buf = torch.zeros(8, dtype=torch.int64, device='cuda:0')
        lib = _lib.load()
        lib.osa_debug_set_rollout_clock_buffer(_lib.ptr(buf))
        one()
        torch.cuda.synchronize()
        lib.osa_debug_set_rollout_clock_buffer(None)
        names = ('policy_step', 'env_step', 'partial_moments', 'grid_barrier', 'merge', 'normalise', 'bootstrap_values',
                 'episode_accounting')
        clocks = {k: round(v / 100.0, 2) for k, v in zip(names, buf.tolist())}
        clocks['sum'] = round(sum(buf.tolist()) / 100.0, 2)
    wall, dev = [], []
    for _ in range(args.epochs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        one()
        e1.record()
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) * 1e3)
        dev.append(e0.elapsed_time(e1))
    wall.sort()
    dev.sort()
    return {'mode': mode, 'path': ad.last_rollout_path if not getattr(ad, 'last_rollout_graphed', False) else 'graph',
            'wall_ms_median': round(wall[len(wall) // 2], 4), 'wall_ms_min': round(wall[0], 4),
            'device_ms_median': round(dev[len(dev) // 2], 4), 'device_ms_min': round(dev[0], 4),
            'us_per_vector_step': round(dev[len(dev) // 2] * 1e3 / T, 2), 'workgroup0_phase_us_per_epoch': clocks}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=4096)
    ap.add_argument('--steps-per-env', type=int, default=16)
    ap.add_argument('--epochs', type=int, default=30)
    args = ap.parse_args()
    from omnisafe_amd import build

    out = {'envs': args.envs, 'steps_per_env': args.steps_per_env, 'epochs': args.epochs,
           '_abi_digest': build.source_digest(),
           'note': 'device_ms: events around adapter.rollout (rollout kernels + get() prefetch + episode flush)',
           'modes': [run(m, args) for m in ('persistent', 'persistent-critics-inside', 'graph', 'launches')]}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
