"""GPU tool: wall-clock of the rollout (T=16 steps, N=4096) vs the device time of its kernels."""
import os, sys, time, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omnisafe_amd
cfg = {'seed': 0, 'train_cfgs': {'device': 'cuda:0', 'total_steps': 65536 * 100, 'vector_env_nums': 4096},
       'algo_cfgs': {'steps_per_epoch': 65536, 'update_iters': 1},
       'logger_cfgs': {'log_dir': tempfile.mkdtemp(), 'verbose': False}}
algo = omnisafe_amd.Agent('PPOLag', 'SynthPointGoal1-v0', custom_cfgs=cfg).agent
def roll():
    algo._env.rollout(steps_per_epoch=algo._steps_per_epoch, agent=algo._actor_critic, buffer=algo._buf, logger=algo._logger)
    algo._buf.ptr = 0
for _ in range(3): roll()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): roll()
torch.cuda.synchronize()
print('rollout wall-clock per epoch (16 steps): %.2f ms' % ((time.perf_counter() - t0) / 10 * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): roll()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
