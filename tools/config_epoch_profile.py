#!/usr/bin/env python
"""A few epochs of one BASELINE config (same construction as tools/baseline_configs.py) for rocprofv3:

    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg3 -- \\
        python $R/tools/config_epoch_profile.py 3
"""
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import omnisafe_amd  # noqa: E402

CONFIGS = {'2': ('PPOLag', 'SynthPointGoal1-v0'), '3': ('CPO', 'SynthCarGoal1-v0'),
           '4': ('PPOLag', 'SynthHumanoid-v0'), '5': ('TRPOLag', 'SynthAnt-v0')}
algo, env_id = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else '3']
EPOCHS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096  # BASELINE config 1: 4 envs x 5000 steps (PPOLag.yaml defaults)
T = int(sys.argv[4]) if len(sys.argv) > 4 else 16
cfg = {'seed': 0, 'train_cfgs': {'device': 'cuda:0', 'vector_env_nums': N, 'total_steps': N * T * (EPOCHS + 1)},
       'algo_cfgs': {'steps_per_epoch': N * T},
       'logger_cfgs': {'log_dir': tempfile.mkdtemp(), 'save_model_freq': 10 ** 9, 'verbose': False},
       'env_cfgs': {'horizon': min(T, 1000), 'cost_p': 0.05}}
if algo == 'PPOLag':
    cfg['algo_cfgs']['kl_early_stop'] = False
a = omnisafe_amd.Agent(algo, env_id, custom_cfgs=cfg).agent
for e in range(EPOCHS):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a._env.rollout(steps_per_epoch=a._steps_per_epoch, agent=a._actor_critic, buffer=a._buf, logger=a._logger)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    a._update()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    a._logger.dump_tabular()
    print(f'epoch {e}: rollout {1e3 * (t1 - t0):.2f} ms, update {1e3 * (t2 - t1):.2f} ms -> {N * T / (t2 - t0):.0f} env-steps/s '
          f'(rollout graphed: {getattr(a._env, "last_rollout_graphed", None)})', flush=True)
# explicit teardown while the runtime (and a profiler wrapped around it) is still up: the rollout's hipGraph and
# the device buffers are released here, not by finalisers at interpreter exit
a._logger.close()
a._env.close()
del a
import gc  # noqa: E402

gc.collect()
torch.cuda.synchronize()
