"""GPU tool (round 4): what the host-env bridge costs per vector step with a ZERO-COST host env -- the path a
Safety-Gymnasium user gets (north_star: "feeds thousands of vectorized Safety-Gymnasium envs"; the reference spends
0.54 s per vector step at N = 4096 on its per-env Python loop).  The host env returns preallocated arrays
of the BASELINE.md section-2 synthetic shapes (obs 60, act 2; truncation every `horizon` steps), so everything
measured is the bridge and the device work around it:

  wait_device_and_d2h       host blocked on the action: policy step of this vector step + the 4 D_a-byte copy
  host_env_step             the (empty) env.step call
  staging_and_h2d_enqueue   host copies into the pinned staging block + enqueueing ONE upload
  rest                      host enqueueing of the device kernels (normaliser, bootstrap values, accounting)

with the previous step's accounting deferred behind the copy (OSA_HOST_DEFER=1, default) and in line (=0).

    python tools/host_env_bridge_timing.py [--out gpurun_out/r4_host_env_bridge.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omnisafe_amd  # noqa: E402
from omnisafe_amd import envs as envs_mod  # noqa: E402
from omnisafe_amd.host_env import HostEnvBridge  # noqa: E402
from omnisafe_amd.spaces import Box  # noqa: E402


class NullHostEnv:
    """Reference env interface (envs/core.py:37-182) with no work inside: fixed observation / reward / cost arrays."""

    need_auto_reset_wrapper = False
    need_time_limit_wrapper = False
    need_evaluation = False

    def __init__(self, num_envs, horizon, obs_dim=60, act_dim=2):
        self.num_envs, self._h, self._t = num_envs, horizon, 0
        self.observation_space = Box(-np.inf, np.inf, (obs_dim,))
        self.action_space = Box(-1.0, 1.0, (act_dim,))
        rng = np.random.default_rng(0)
        self._obs = torch.from_numpy(rng.standard_normal((num_envs, obs_dim)).astype(np.float32))
        self._r = torch.from_numpy(rng.standard_normal(num_envs).astype(np.float32))
        self._c = torch.from_numpy((rng.random(num_envs) < 0.05).astype(np.float32))
        self._f = torch.zeros(num_envs, dtype=torch.bool)
        self._tr = torch.ones(num_envs, dtype=torch.bool)
        self.env_spec_log = {}
        self.max_episode_steps = horizon

    def set_seed(self, seed):
        pass

    def reset(self, seed=None, options=None):
        self._t = 0
        return self._obs, {}

    def step(self, action):
        self._t += 1
        if self._t >= self._h:
            self._t = 0
            return self._obs, self._r, self._c, self._f, self._tr, {'final_observation': self._obs,
                                                                     '_final_observation': self._tr}
        return self._obs, self._r, self._c, self._f, self._f, {}

    def close(self):
        pass


def run(N, T, defer):
    os.environ['OSA_HOST_DEFER'] = '1' if defer else '0'
    envs_mod.ENV_REGISTRY['HostNull-v0'] = lambda env_id, num_envs=1, device='cuda:0', **kw: HostEnvBridge(
        NullHostEnv(num_envs, kw.get('horizon', T)), device)
    import tempfile

    cfg = {'seed': 0, 'train_cfgs': {'device': 'cuda:0', 'total_steps': 4 * N * T, 'vector_env_nums': N},
           'algo_cfgs': {'steps_per_epoch': N * T, 'update_iters': 1, 'batch_size': 16384},
           'logger_cfgs': {'log_dir': tempfile.mkdtemp(prefix='osa_host_'), 'verbose': False}, 'env_cfgs': {'horizon': T}}
    algo = omnisafe_amd.Agent('PPOLag', 'HostNull-v0', custom_cfgs=cfg).agent
    bridge = algo._env._env
    assert isinstance(bridge, HostEnvBridge)
    for _ in range(2):  # warm-up
        algo._buf.ptr = 0
        algo._env.rollout(steps_per_epoch=T, agent=algo._actor_critic, buffer=algo._buf, logger=algo._logger)
    torch.cuda.synchronize()
    bridge.timing = {}
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        algo._buf.ptr = 0
        algo._env.rollout(steps_per_epoch=T, agent=algo._actor_critic, buffer=algo._buf, logger=algo._logger)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    steps = reps * T
    rec = {'N': N, 'T': T, 'deferred_accounting': bool(defer), 'us_per_vector_step': round(wall / steps * 1e6, 1)}
    for k, v in bridge.timing.items():
        rec['us_' + k] = round(v / steps * 1e6, 1)
    rec['us_rest_host_enqueue_and_epoch_ends'] = round(rec['us_per_vector_step'] - sum(
        rec['us_' + k] for k in bridge.timing), 1)
    down, up = bridge.pcie_bytes_per_env_step()
    rec['pcie_bytes_per_env_step_down_up'] = [round(down, 1), round(up, 1)]
    rec['env_steps_per_s_bridge_bound'] = round(N * steps / wall, 0)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    rows = []
    for N, T in ((4096, 16), (4096, 64), (256, 64), (16, 256)):
        for defer in (1, 0):
            rows.append(run(N, T, defer))
            print(json.dumps(rows[-1]), flush=True)
    if args.out:
        json.dump({'device': torch.cuda.get_device_name(0), 'host_cpus': os.cpu_count(), 'rows': rows},
                  open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
