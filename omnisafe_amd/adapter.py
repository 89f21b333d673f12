"""Device-resident OnPolicyAdapter.

Mirror of omnisafe/adapter/onpolicy_adapter.py:30-190 (+ the wrapper chain of online_adapter.py:85-140):
``OnPolicyAdapter(env_id, num_envs, seed, cfgs)`` and ``rollout(steps_per_epoch, agent, buffer,
logger)``.  The reference walks all N envs in Python at every step (2N tensor->bool host syncs, N x 7
index assignments into N buffers, a single-row ``agent.step`` per finished env); here one vector step
is a short, sync-free sequence of kernels:

    osa_policy_step          pi / V_r / V_c forward + sample + logp, written straight into buffer row t
    osa_action_scale         ActionScale wrapper (envs/wrapper.py:510-514)
    env.step                 (synthetic env: osa_synth_env_step)
    osa_normalizer_push/apply   ObsNormalize wrapper (final_observation rows first, then next obs,
                             envs/wrapper.py:231-241), result written into buffer row t+1
    osa_policy_step (critics)   bootstrap values V(final_obs) / V(next_obs), batched
    osa_rollout_post_step    episode accounting + bootstrap selection -> path_end / boot rows

Episode metrics are extracted from the device once per epoch and appended to the logger in the
reference's (step, env) order, so the window-100 statistics and the Lagrange update see the same data.
"""
from __future__ import annotations

import torch

from . import _lib
from . import envs as envs_mod
from .buffer import VectorOnPolicyBuffer
from .models import ConstraintActorCritic
from .normalizer import Normalizer


class OnPolicyAdapter:  # pylint: disable=too-many-instance-attributes
    def __init__(self, env_id: str, num_envs: int, seed: int, cfgs, env=None) -> None:
        self._lib = _lib.load(require_gpu=True)
        self._cfgs = cfgs
        self._env_id = env_id
        self._device = torch.device(cfgs.train_cfgs.device)
        env_cfgs = {}
        if hasattr(cfgs, 'env_cfgs') and cfgs.env_cfgs is not None:
            env_cfgs = cfgs.env_cfgs.todict() if hasattr(cfgs.env_cfgs, 'todict') else dict(cfgs.env_cfgs)
        self._env = env if env is not None else envs_mod.make(env_id, num_envs=num_envs,
                                                               device=self._device, **env_cfgs)
        if getattr(self._env, 'need_auto_reset_wrapper', False) or getattr(
                self._env, 'need_time_limit_wrapper', False):
            raise NotImplementedError('TimeLimit/AutoReset wrappers are not on the accelerated path: '
                                      'the env must auto-reset (gymnasium vector convention)')
        a = cfgs.algo_cfgs
        self._num_envs = int(self._env.num_envs)
        self._obs_dim = int(self._env.observation_space.shape[0])
        self._act_dim = int(self._env.action_space.shape[0])
        N, dev = self._num_envs, self._device
        f32 = dict(dtype=torch.float32, device=dev)
        self._obs_normalizer = (Normalizer((self._obs_dim,), clip=5, device=dev)
                                if a.obs_normalize else None)
        # RewardNormalize / CostNormalize wrappers (envs/wrapper.py:280-423): scalar running statistics
        # over the batch of N rewards (costs) of each step, clip 5; episode metrics keep the ORIGINAL
        # reward / cost (info['original_reward'], onpolicy_adapter.py:155-156)
        self._reward_normalizer = (Normalizer((1,), clip=5, device=dev)
                                   if getattr(a, 'reward_normalize', False) else None)
        self._cost_normalizer = (Normalizer((1,), clip=5, device=dev)
                                 if getattr(a, 'cost_normalize', False) else None)
        # ActionScale(low=-1, high=1): agent acts in [-1, 1], env receives [space.low, space.high]
        self._old_min = torch.as_tensor(self._env.action_space.low, **f32).reshape(-1).contiguous()
        self._old_max = torch.as_tensor(self._env.action_space.high, **f32).reshape(-1).contiguous()
        self._act_env = torch.empty(N, self._act_dim, **f32)
        self._last_obs = torch.empty(N, self._obs_dim, **f32)
        self._final_norm = torch.empty(N, self._obs_dim, **f32)
        self._ep_ret = torch.zeros(N, **f32)
        self._ep_cost = torch.zeros(N, **f32)
        self._ep_len = torch.zeros(N, **f32)
        self._ep_rows: dict[str, torch.Tensor] = {}
        self._env.set_seed(seed)
        self._seed = seed

    # ------------------------------------------------------------------ reference surface
    @property
    def observation_space(self):
        return self._env.observation_space

    @property
    def action_space(self):
        from .spaces import Box

        return Box(-1.0, 1.0, (self._act_dim,))  # ActionScale's rescaled space

    @property
    def num_envs(self) -> int:
        return self._num_envs

    @property
    def env_spec_keys(self) -> list[str]:
        return list(getattr(self._env, 'env_spec_log', {}) or [])

    def save(self) -> dict:
        """online_adapter.py:232-246: objects to checkpoint."""
        saved = {}
        if self._obs_normalizer is not None:
            saved['obs_normalizer'] = self._obs_normalizer
        if self._reward_normalizer is not None:
            saved['reward_normalizer'] = self._reward_normalizer
        if self._cost_normalizer is not None:
            saved['cost_normalizer'] = self._cost_normalizer
        return saved

    def close(self) -> None:
        self._env.close()

    def reset(self, seed: int | None = None):
        obs, info = self._env.reset(seed=seed)
        return self._normalize(obs), info

    def _normalize(self, raw: torch.Tensor, mask: torch.Tensor | None = None,
                   out: torch.Tensor | None = None) -> torch.Tensor:
        raw = raw.reshape(self._num_envs, self._obs_dim)
        if self._obs_normalizer is None:
            if out is None:
                return raw
            out.copy_(raw)
            return out
        return self._obs_normalizer.normalize(raw, mask=mask, out=out)

    def _reset_log(self) -> None:
        self._ep_ret.zero_()
        self._ep_cost.zero_()
        self._ep_len.zero_()

    def _ensure_episode_rows(self, T: int) -> None:
        if not self._ep_rows or self._ep_rows['done'].shape[0] != T:
            N, dev = self._num_envs, self._device
            self._ep_rows = {'done': torch.zeros(T, N, dtype=torch.uint8, device=dev)}
            for k in ('ret', 'cost', 'len'):
                self._ep_rows[k] = torch.zeros(T, N, dtype=torch.float32, device=dev)

    # ------------------------------------------------------------------ rollout
    def rollout(self, steps_per_epoch: int, agent: ConstraintActorCritic, buffer: VectorOnPolicyBuffer,
                logger) -> None:
        """onpolicy_adapter.py:58-136."""
        lib, N, T = self._lib, self._num_envs, int(steps_per_epoch)
        assert buffer.size == T and buffer.num_buffers == N and buffer.ptr == 0
        self._reset_log()
        self._ensure_episode_rows(T)
        ep = self._ep_rows
        ep['done'].zero_()
        b = buffer.data
        obs_raw, _ = self._env.reset()  # the reference resets every epoch (:80)
        self._normalize(obs_raw, out=b['obs'][0])
        for t in range(T):
            st = _lib.stream_ptr()
            obs = b['obs'][t]
            agent.step(obs, out={'act': b['act'][t], 'value_r': b['value_r'][t], 'value_c': b['value_c'][t],
                                 'logp': b['logp'][t]})
            _lib.check(lib.osa_action_scale(_lib.ptr(b['act'][t]), self._act_dim, _lib.ptr(self._act_env),
                                            self._act_dim, N, self._act_dim, _lib.ptr(self._old_min),
                                            _lib.ptr(self._old_max), -1.0, 1.0, st), 'osa_action_scale')
            next_raw, reward, cost, terminated, truncated, info = self._env.step(self._act_env)
            reward, cost = reward.reshape(N), cost.reshape(N)
            if self._reward_normalizer is not None:
                self._reward_normalizer.normalize(reward.reshape(N, 1), out=b['reward'][t].view(N, 1))
            else:
                b['reward'][t].copy_(reward)
            if self._cost_normalizer is not None:
                self._cost_normalizer.normalize(cost.reshape(N, 1), out=b['cost'][t].view(N, 1))
            else:
                b['cost'][t].copy_(cost)
            reward = reward.to(torch.float32).contiguous()  # original values for the episode metrics
            cost = cost.to(torch.float32).contiguous()
            term = terminated.reshape(N).to(torch.uint8)
            trunc = truncated.reshape(N).to(torch.uint8)
            vfinal = (None, None)
            if 'final_observation' in info:
                fmask = info.get('_final_observation', None)
                fmask = (term | trunc) if fmask is None else fmask.reshape(N).to(torch.uint8)
                self._normalize(info['final_observation'], mask=fmask, out=self._final_norm)
                vfinal = agent.values(self._final_norm)
            epoch_end = t >= T - 1
            nxt = self._last_obs if epoch_end else b['obs'][t + 1]
            self._normalize(next_raw, out=nxt)
            vnext = agent.values(nxt) if epoch_end else (None, None)
            _lib.check(lib.osa_rollout_post_step(
                N, int(epoch_end), _lib.ptr(reward), _lib.ptr(cost), _lib.ptr(term),
                _lib.ptr(trunc), _lib.ptr(vnext[0]), _lib.ptr(vnext[1]), _lib.ptr(vfinal[0]),
                _lib.ptr(vfinal[1]), _lib.ptr(self._ep_ret), _lib.ptr(self._ep_cost),
                _lib.ptr(self._ep_len), _lib.ptr(b['path_end'][t]), _lib.ptr(b['boot_r'][t]),
                _lib.ptr(b['boot_c'][t]), _lib.ptr(ep['done'][t]), _lib.ptr(ep['ret'][t]),
                _lib.ptr(ep['cost'][t]), _lib.ptr(ep['len'][t]), st), 'osa_rollout_post_step')
            buffer.advance()
        self._flush_logs(logger, buffer)

    def _flush_logs(self, logger, buffer: VectorOnPolicyBuffer) -> None:
        """One device->host transfer per epoch: finished episodes in (step, env) order
        (_log_metrics, :159-174) and the mean critic outputs (logger.store Value/*, :88-92)."""
        ep = self._ep_rows
        window = 100
        if hasattr(logger, '_headers_windows'):
            window = logger._headers_windows.get('Metrics/EpRet') or 100  # noqa: SLF001
        idx = ep['done'].reshape(-1).nonzero().reshape(-1)  # host sync (once per epoch)
        if idx.numel() > 0:
            idx = idx[-window:]
            vals = torch.stack([ep[k].reshape(-1)[idx] for k in ('ret', 'cost', 'len')]).cpu()
            logger.extend('Metrics/EpRet', vals[0].tolist())
            logger.extend('Metrics/EpCost', vals[1].tolist())
            logger.extend('Metrics/EpLen', vals[2].tolist())
        logger.store({'Value/reward': float(buffer.data['value_r'].mean())})
        if self._cfgs.algo_cfgs.use_cost:
            logger.store({'Value/cost': float(buffer.data['value_c'].mean())})
