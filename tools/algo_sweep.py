"""GPU tool: env-steps/s of every accelerated algorithm with ITS OWN YAML defaults (batch size, passes) on
the headline shapes (SafetyPointGoal1: obs 60, act 2; 4096 envs x 16 steps = 65 536 env-steps per epoch,
kl_early_stop off so that every epoch does the maximum work).  Prints a markdown table."""
import os, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omnisafe_amd
from omnisafe_amd import config

ALGOS = (sys.argv[1].split(',') if len(sys.argv) > 1 else None) or ['PolicyGradient', 'PPO', 'PPOLag', 'PDO', 'IPO', 'CPPOPID', 'P3O', 'FOCOPS', 'CUP', 'PPOSaute',
         'PPOSimmerPID', 'NaturalPG', 'TRPO', 'TRPOLag', 'RCPO', 'TRPOPID', 'OnCRPO', 'CPO', 'PCPO', 'TRPOSaute',
         'TRPOSimmerPID']
N, T, EPOCHS, WARM = 4096, 16, 3, 3
rows = []
for algo in ALGOS:
    d = config.get_default_kwargs(algo)
    extra = {'steps_per_epoch': N * T, 'kl_early_stop': False}
    if 'max_ep_len' in d['algo_cfgs']:
        extra['max_ep_len'] = 16
    cfg = {'seed': 0, 'train_cfgs': {'device': 'cuda:0', 'vector_env_nums': N, 'total_steps': N * T * (EPOCHS + WARM + 1)},
           'algo_cfgs': extra, 'logger_cfgs': {'log_dir': tempfile.mkdtemp(), 'save_model_freq': 10 ** 9, 'verbose': False},
           'env_cfgs': {'horizon': 16, 'cost_p': 0.05}}
    a = omnisafe_amd.Agent(algo, 'SynthPointGoal1-v0', custom_cfgs=cfg).agent

    def epoch():
        a._env.rollout(steps_per_epoch=a._steps_per_epoch, agent=a._actor_critic, buffer=a._buf, logger=a._logger)
        a._update()
        a._logger.dump_tabular()

    for _ in range(WARM):
        epoch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(EPOCHS):
        epoch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / EPOCHS
    ac = d['algo_cfgs']
    rows.append((algo, ac['batch_size'], ac['update_iters'], dt * 1e3, N * T / dt))
    print(f'{algo:14s} B={ac["batch_size"]:4d} iters={ac["update_iters"]:3d}  {dt * 1e3:8.1f} ms/epoch  {N * T / dt:10.0f} env-steps/s', flush=True)
print('\n| algorithm | batch_size | update_iters | ms / epoch | env-steps/s |\n|---|---|---|---|---|')
for r in rows:
    print(f'| {r[0]} | {r[1]} | {r[2]} | {r[3]:.1f} | {r[4]:,.0f} |')
