"""TEST / MEASUREMENT INFRASTRUCTURE -- times the UNMODIFIED reference on the host CPU (bench.py's
`cpu_baseline` leg, kind "reference"; SURVEY.md 8d "CPU baseline").

    python oracle/ref_cpu_baseline.py --envs 4096 --steps-per-env 16 --batch-size 64 --update-iters 40 \
        --sample-iters 2 --threads 16

Runs `omnisafe.Agent('PPOLag', 'SynthPointGoal1-v0', custom_cfgs={device: 'cpu', ...}).learn()` -- the
reference's own facade, adapter, buffer and update loop (algo_wrapper.py:167-184,
policy_gradient.py:238-306) on the CPU twin of the synthetic env (oracle/ref_harness.py) -- for ONE epoch
of the benchmark's workload shape with `update_iters` lowered to `--sample-iters` so that it fits the time
budget of a default bench run, then reads `Time/Rollout`, `Time/Update`, `Time/FPS` from its progress.csv
(policy_gradient.py:279-282) and scales the update to the full number of passes (every pass is the same
work: ceil(M/B) minibatch triples + one KL pass).  Prints ONE JSON line.

The reference comes from /root/reference (build container) or from oracle/_ref/omnisafe_ref.zip (GPU box;
staged by oracle/stage_reference.py).  Runs in its own process: the import stubs for gymnasium & co and the
torch thread count stay out of the benchmark process.
"""
from __future__ import annotations

import argparse
import contextlib
import csv
import glob
import io
import json
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument('--algo', default='PPOLag')
    ap.add_argument('--env-id', default='SynthPointGoal1-v0')
    ap.add_argument('--envs', type=int, default=4096)
    ap.add_argument('--steps-per-env', type=int, default=16)
    ap.add_argument('--batch-size', type=int, default=64)
    ap.add_argument('--update-iters', type=int, default=40)
    ap.add_argument('--sample-iters', type=int, default=2)
    ap.add_argument('--threads', type=int, default=16)
    args = ap.parse_args()
    if args.sample_iters <= 0:
        args.sample_iters = args.update_iters
    import ref_harness

    if not ref_harness.reference_available():
        print(json.dumps({'error': 'reference not available'}))
        return 1
    omnisafe = ref_harness.import_reference()
    ref_harness.register_synth_env()
    from omnisafe.utils.config import get_default_kwargs_yaml

    has_env_cfgs = 'env_cfgs' in get_default_kwargs_yaml(args.algo, args.env_id, 'on-policy').todict()
    ref_harness.DEFAULT_HORIZON = args.steps_per_env  # TRPOLag.yaml / CPO.yaml have no env_cfgs key to carry it
    spe = args.envs * args.steps_per_env
    log_dir = tempfile.mkdtemp(prefix='osa_refbase_')
    cfg = {'seed': 0,
           'train_cfgs': {'device': 'cpu', 'torch_threads': args.threads, 'vector_env_nums': args.envs,
                          'total_steps': spe, 'parallel': 1},
           'algo_cfgs': {'steps_per_epoch': spe, 'batch_size': args.batch_size,
                         'update_iters': args.sample_iters},
           'logger_cfgs': {'use_wandb': False, 'use_tensorboard': False, 'log_dir': log_dir,
                           'save_model_freq': 10 ** 9}}
    if has_env_cfgs:
        cfg['env_cfgs'] = {'horizon': args.steps_per_env, 'cost_p': 0.05}
    if args.batch_size <= 0:  # keep the algorithm's YAML default
        del cfg['algo_cfgs']['batch_size']
    if 'kl_early_stop' in get_default_kwargs_yaml(args.algo, args.env_id, 'on-policy').todict()['algo_cfgs']:
        cfg['algo_cfgs']['kl_early_stop'] = False
    sink = io.StringIO()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
        agent = omnisafe.Agent(args.algo, args.env_id, custom_cfgs=cfg)
        t_init = time.perf_counter() - t0
        # buffer.get() (finish + concat + standardise) runs once per epoch, not once per pass: time it from
        # outside (a wrapper around the bound method, the reference's code is untouched) so that only the
        # per-pass part of Time/Update is scaled
        buf, t_get = agent.agent._buf, [0.0]  # noqa: SLF001
        orig_get = buf.get

        def timed_get():
            t = time.perf_counter()
            out = orig_get()
            t_get[0] += time.perf_counter() - t
            return out

        buf.get = timed_get
        agent.agent.learn()  # BaseAlgo.learn: the facade's learn() would also build a Plotter (not on the path)
    wall = time.perf_counter() - t0
    rows = list(csv.DictReader(open(glob.glob(os.path.join(log_dir, '**', 'progress.csv'), recursive=True)[0])))
    r = rows[-1]
    t_roll, t_upd, fps = float(r['Time/Rollout']), float(r['Time/Update']), float(r['Time/FPS'])
    t_passes = t_upd - t_get[0]
    t_epoch_full = t_roll + t_get[0] + t_passes * args.update_iters / args.sample_iters
    import torch

    print(json.dumps({
        'value': round(spe / t_epoch_full, 1), 'unit': 'env-steps/s', 'cores': torch.get_num_threads(),
        'kind': 'reference',
        'sample': (f'unmodified omnisafe.Agent({args.algo!r}, device=cpu, torch_threads={args.threads}).learn(), '
                   f'1 epoch of the benchmark shape ({args.envs} envs x {args.steps_per_env} steps = {spe} '
                   f'env-steps, batch_size {args.batch_size if args.batch_size > 0 else "= YAML default"}) with '
                   f'update_iters={args.sample_iters}: '
                   f'Time/Rollout {t_roll:.2f} s, Time/Update {t_upd:.2f} s, Time/FPS {fps:.0f} (its own csv); '
                   f'buffer.get() {t_get[0]:.2f} s of the update; value = {spe} / (Time/Rollout + get + '
                   f'(Time/Update - get) x {args.update_iters}/{args.sample_iters}), i.e. only the minibatch '
                   f'passes scaled to the benchmark\'s {args.update_iters}'),
        'time_rollout_s': round(t_roll, 3), 'time_update_s': round(t_upd, 3), 'time_get_s': round(t_get[0], 3),
        'measured_fps_at_sample_iters': round(fps, 1), 'sample_iters': args.sample_iters,
        'wall_s': round(wall, 2), 'init_s': round(t_init, 2), 'host_cpus': os.cpu_count()}))
    return 0


if __name__ == '__main__':
    sys.exit(main())
