#!/usr/bin/env python
"""Single-GPU env-steps/s (rollout + update, = the reference's Time/FPS) of BASELINE.json configs 2-5 at their
observation / action shapes, 4096 device envs x 16 steps per epoch, each algorithm's YAML defaults
(kl_early_stop off = maximum work), next to the UNMODIFIED reference on this box's host cores
(oracle/ref_cpu_baseline.py).  -> gpurun_out/r4_baseline_configs.{json,md}

    python tools/baseline_configs.py [--no-reference]
"""
import json
import os
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import omnisafe_amd  # noqa: E402

CONFIGS = [('2', 'PPOLag', 'SynthPointGoal1-v0', 'SafetyPointGoal1 60/2', 40, 2),
           ('3', 'CPO', 'SynthCarGoal1-v0', 'SafetyCarGoal1 72/2', 10, 0),
           ('4', 'PPOLag', 'SynthHumanoid-v0', 'SafetyHumanoidVelocity 376/17', 40, 1),
           ('5', 'TRPOLag', 'SynthAnt-v0', 'SafetyAntVelocity 27/8', 10, 0)]
N, T, WARM, STEPS = 4096, 16, 3, 5
rows = []
for tag, algo, env_id, label, iters, ref_sample in CONFIGS:
    cfg = {'seed': 0, 'train_cfgs': {'device': 'cuda:0', 'vector_env_nums': N, 'total_steps': N * T * (WARM + STEPS + 1)},
           'algo_cfgs': {'steps_per_epoch': N * T},
           'logger_cfgs': {'log_dir': tempfile.mkdtemp(), 'save_model_freq': 10 ** 9, 'verbose': False},
           'env_cfgs': {'horizon': T, 'cost_p': 0.05}}
    if algo == 'PPOLag':
        cfg['algo_cfgs']['kl_early_stop'] = False
    a = omnisafe_amd.Agent(algo, env_id, custom_cfgs=cfg).agent

    def epochs(n):
        for _ in range(n):
            a._env.rollout(steps_per_epoch=a._steps_per_epoch, agent=a._actor_critic, buffer=a._buf, logger=a._logger)
            a._update()
            a._logger.dump_tabular()
        torch.cuda.synchronize()

    epochs(WARM)
    t0 = time.perf_counter()
    epochs(STEPS)
    dt = (time.perf_counter() - t0) / STEPS
    row = {'config': tag, 'algo': algo, 'shape': label, 'ms_per_epoch': round(dt * 1e3, 2),
           'gpu_env_steps_per_s': round(N * T / dt, 1), 'update_path': getattr(a._updater, 'last_path', None)}
    if '--no-reference' not in sys.argv:
        threads = max(1, min(os.cpu_count() or 1, 16))
        cmd = [sys.executable, os.path.join(ROOT, 'oracle', 'ref_cpu_baseline.py'), '--algo', algo, '--env-id', env_id,
               '--envs', str(N), '--steps-per-env', str(T), '--batch-size', '0', '--update-iters', str(iters),
               '--sample-iters', str(ref_sample), '--threads', str(threads)]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=1200,
                           env=dict(os.environ, OMP_NUM_THREADS=str(threads)))
        line = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
        ref = json.loads(line[-1]) if line else {'error': p.stderr[-300:]}
        row['reference_cpu'] = ref
        if 'value' in ref:
            row['gpu_over_reference_cpu'] = round(row['gpu_env_steps_per_s'] / ref['value'], 1)
    rows.append(row)
    print(json.dumps(row), flush=True)
    del a
    torch.cuda.empty_cache()
out = os.path.join(ROOT, 'gpurun_out', 'r4_baseline_configs')
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(rows, open(out + '.json', 'w'), indent=1)
with open(out + '.md', 'w') as f:
    f.write('| BASELINE config | algorithm, shapes | MI355X x1 env-steps/s (ms / epoch of 65 536 steps) | update path | '
            'unmodified reference, host CPU env-steps/s (threads) | ratio |\n|---|---|---|---|---|---|\n')
    for r in rows:
        ref = r.get('reference_cpu', {})
        f.write(f"| {r['config']} | {r['algo']}, {r['shape']} | {r['gpu_env_steps_per_s']:,.0f} ({r['ms_per_epoch']}) | "
                f"{r['update_path']} | {ref.get('value', 'n/a')} ({ref.get('cores', '')}) | "
                f"{r.get('gpu_over_reference_cpu', '')} |\n")
print(open(out + '.md').read())
