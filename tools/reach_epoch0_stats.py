"""Large-sample episode statistics of SynthReach-v0 under the INITIAL policy (seed-0 weights): one
rollout of 4096 envs x 300 steps = 24 576 episodes on the device env, for comparison with the same
quantity of the unmodified reference on its CPU twin (standard error ~0.05 on the episode cost).
Separates rollout-level differences from differences that only appear through learning."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')]
import numpy as np  # noqa: E402

import omnisafe_amd  # noqa: E402
from test_learning_gpu import GOLDEN, reach_custom_cfgs  # noqa: E402

g = json.load(open(GOLDEN))
out = {}
for algo_name in sys.argv[1:] or ['PPOLag', 'PPOSaute']:
    N, T = 4096, 300
    cfg = dict(g['config'], vector_env_nums=N, steps_per_epoch=N * T, epochs=1)
    custom = reach_custom_cfgs(algo_name, 0, cfg, tempfile.mkdtemp())
    custom['logger_cfgs'].update({'window_lens': 100000, 'verbose': False})
    algo = omnisafe_amd.Agent(algo_name, 'SynthReach-v0', custom_cfgs=custom).agent
    algo._env.rollout(steps_per_epoch=T, agent=algo._actor_critic, buffer=algo._buf, logger=algo._logger)
    c = np.asarray(list(algo._logger._data['Metrics/EpCost']), np.float64)
    r = np.asarray(list(algo._logger._data['Metrics/EpRet']), np.float64)
    out[algo_name] = {'episodes': len(c), 'EpCost': c.mean(), 'EpCost_se': c.std() / np.sqrt(len(c)),
                      'EpRet': r.mean(), 'EpRet_se': r.std() / np.sqrt(len(r)), 'EpCost_std': c.std(),
                      'frac_cost0': float((c == 0).mean())}
    print(algo_name, out[algo_name])
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'reach_epoch0_stats.json'), 'w'), indent=1)
