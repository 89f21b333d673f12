#!/usr/bin/env python
"""TEST INFRASTRUCTURE (not product code).  Per-kernel CPU baselines of the UNMODIFIED reference on this host
("per-kernel baselines"): the reference's own code for what the HIP kernels replace --

  * `VectorOnPolicyBuffer.store` x T, `finish_path` x N, `get()`         (-> osa_gae_scan, osa_adv_stats_*, osa_buffer_get)
  * one minibatch step of `PolicyGradient._update` for PPOLag (batch 64)  (-> one step of osa_ppo_pass)
  * `NaturalPG._fvp` on the full batch                                    (-> osa_actor_fvp_raw + osa_fvp_finish)
  * `ConstraintActorCritic.step` on N observations                        (-> osa_policy_step)

at BASELINE config 2's shapes (N = 4096 envs x T = 16 steps, obs 60, act 2).  Prints one JSON.

    OMP_NUM_THREADS=8 python oracle/ref_kernel_baselines.py
"""
import json
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

ref_harness.import_reference()
import torch  # noqa: E402

threads = int(os.environ.get('OMP_NUM_THREADS', os.cpu_count() or 1))
torch.set_num_threads(threads)
import omnisafe  # noqa: E402
from omnisafe.common.buffer import VectorOnPolicyBuffer  # noqa: E402

ref_harness.register_synth_env()
N, T, D_O, D_A = 4096, 16, 60, 2
res = {'host_threads': threads, 'shape': f'N={N} T={T} obs={D_O} act={D_A}'}


def timed(fn, reps=1):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


# ---- buffer: store x T, finish_path x N, get()
from gymnasium.spaces import Box  # noqa: E402  (the harness' stub Box)
import numpy as np  # noqa: E402

obs_space, act_space = Box(-np.inf, np.inf, (D_O,)), Box(-1.0, 1.0, (D_A,))
buf = VectorOnPolicyBuffer(obs_space, act_space, size=T, gamma=0.99, lam=0.95, lam_c=0.95, advantage_estimator='gae',
                           penalty_coefficient=0.0, standardized_adv_r=True, standardized_adv_c=True, num_envs=N,
                           device=torch.device('cpu'))
rows = [dict(obs=torch.randn(N, D_O), act=torch.randn(N, D_A), reward=torch.randn(N), cost=torch.rand(N),
             value_r=torch.randn(N), value_c=torch.randn(N), logp=torch.randn(N)) for _ in range(T)]
t_store = t_finish = t_get = 0.0
for _ in range(2):
    t0 = time.perf_counter()
    for r in rows:
        buf.store(**r)
    t1 = time.perf_counter()
    for i in range(N):
        buf.finish_path(torch.zeros(1), torch.zeros(1), i)
    t2 = time.perf_counter()
    buf.get()
    t3 = time.perf_counter()
    t_store, t_finish, t_get = t1 - t0, t2 - t1, t3 - t2
res['buffer'] = {'store_s': round(t_store, 3), 'finish_path_s': round(t_finish, 3), 'get_s': round(t_get, 3),
                 'transitions': N * T}

# ---- PPOLag: one epoch with update_iters = 1 -> Time/Update / number of minibatches; policy step
d = tempfile.mkdtemp()
cfg = {'seed': 0, 'train_cfgs': {'device': 'cpu', 'torch_threads': threads, 'vector_env_nums': N, 'total_steps': N * T},
       'algo_cfgs': {'steps_per_epoch': N * T, 'update_iters': 1, 'kl_early_stop': False},
       'logger_cfgs': {'log_dir': d, 'use_wandb': False, 'use_tensorboard': False, 'save_model_freq': 10 ** 9},
       'env_cfgs': {'horizon': T, 'cost_p': 0.05}}
agent = omnisafe.Agent('PPOLag', 'SynthPointGoal1-v0', custom_cfgs=cfg)
algo = agent.agent
algo._env.rollout(steps_per_epoch=algo._steps_per_epoch, agent=algo._actor_critic, buffer=algo._buf, logger=algo._logger)
t0 = time.perf_counter()
algo._update()
dt = time.perf_counter() - t0
res['ppolag_minibatch_step'] = {'seconds_per_pass': round(dt, 3), 'steps_per_pass': N * T // 64,
                                'us_per_step': round(dt / (N * T // 64) * 1e6, 1)}
obs = torch.randn(N, D_O)
res['policy_step'] = {'us_per_call': round(timed(lambda: algo._actor_critic.step(obs), 20) * 1e6, 1), 'rows': N}

# ---- TRPOLag: the Fisher-vector product on the full batch
cfg['algo_cfgs'].pop('kl_early_stop')
cfg['algo_cfgs'].pop('update_iters')
cfg.pop('env_cfgs')  # (TRPOLag.yaml has no env_cfgs block: the env's default horizon applies)
cfg['logger_cfgs']['log_dir'] = tempfile.mkdtemp()
tr = omnisafe.Agent('TRPOLag', 'SynthPointGoal1-v0', custom_cfgs=cfg).agent
tr._env.rollout(steps_per_epoch=tr._steps_per_epoch, agent=tr._actor_critic, buffer=tr._buf, logger=tr._logger)
data = tr._buf.get()
tr._fvp_obs = data['obs'][:: tr._cfgs.algo_cfgs.fvp_sample_freq]
from omnisafe.utils.tools import get_flat_params_from  # noqa: E402

theta = get_flat_params_from(tr._actor_critic.actor)
v = torch.randn_like(theta)
res['fvp'] = {'us_per_call': round(timed(lambda: tr._fvp(v), 5) * 1e6, 1), 'rows': int(tr._fvp_obs.shape[0])}
print(json.dumps(res))
