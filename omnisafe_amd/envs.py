"""Environment plug-in boundary (mirror of the part of omnisafe/envs/core.py:37-182,300-421 that the
on-policy adapter touches) and the device-resident synthetic vector CMDP.

An environment handed to :class:`omnisafe_amd.adapter.OnPolicyAdapter` must provide, like a reference
``CMDP``: ``num_envs``, ``observation_space``, ``action_space`` (Box), ``reset(seed=None, options=None)
-> (obs, info)``, ``step(action) -> (obs, reward, cost, terminated, truncated, info)`` returning DEVICE
tensors with gymnasium's vector auto-reset convention (``info['final_observation']`` +
``info['_final_observation']`` on steps where some env finished: envs/wrapper.py:232-238),
``set_seed``, ``close``, ``need_auto_reset_wrapper`` / ``need_time_limit_wrapper`` (must be False here).

Safety-Gymnasium (MuJoCo, CPU, third-party: not in the reference repository, not installed) is outside
this package; ``Synth*`` ids give fixed-shape stand-ins with the observation/action dimensions of the
BASELINE configs.
"""
from __future__ import annotations

from typing import Any, Callable

import numpy as np
import torch

from . import _lib
from .spaces import Box

SYNTH_DIMS = {
    'SynthPointGoal1-v0': (60, 2),     # SafetyPointGoal1-v0 (dims verified from the reference's
                                       # tests/saved_source PPO checkpoint: SURVEY.md section 8)
    'SynthCarGoal1-v0': (72, 2),       # SafetyCarGoal1-v0
    'SynthAnt-v0': (27, 8),            # SafetyAntVelocity-v1
    'SynthHumanoid-v0': (376, 17),     # SafetyHumanoidVelocity-v1
    'SynthTiny-v0': (6, 2),
}

ENV_REGISTRY: dict[str, Callable[..., Any]] = {}


def env_register(cls):
    """Class decorator keyed by ``cls._support_envs`` (envs/core.py:300-360)."""
    for env_id in cls._support_envs:
        ENV_REGISTRY[env_id] = cls
    return cls


def support_envs() -> list[str]:
    return sorted(ENV_REGISTRY)


def make(env_id: str, num_envs: int = 1, device='cuda:0', **env_cfgs):
    """envs/core.py:389-421.  Ids of this package are device-resident envs; any other id is looked up in the
    caller's ``omnisafe`` env registry (Safety-Gymnasium, user classes under ``@env_register``), built on the
    host and driven through :class:`omnisafe_amd.host_env.HostEnvBridge` (one D2H / H2D pair per vector step)."""
    if env_id in ENV_REGISTRY:
        return ENV_REGISTRY[env_id](env_id, num_envs=num_envs, device=device, **env_cfgs)
    from .host_env import make_reference_env

    env = make_reference_env(env_id, num_envs=num_envs, device=device, **env_cfgs)
    if env is None:
        raise KeyError(f'{env_id} is registered neither with omnisafe_amd (known: {support_envs()}) nor with an '
                       'importable omnisafe (omnisafe.envs.core.support_envs())')
    return env


@env_register
class SynthVectorEnv:  # pylint: disable=too-many-instance-attributes
    """Zero-cost synthetic vector CMDP living entirely in HBM (osa_synth_env_step)."""

    _support_envs = list(SYNTH_DIMS)
    need_auto_reset_wrapper = False
    need_time_limit_wrapper = False
    need_evaluation = False

    def __init__(self, env_id: str, num_envs: int = 1, device='cuda:0', horizon: int = 1000,
                 cost_p: float = 0.05, seed: int = 0, **_unused) -> None:
        self._lib = _lib.load(require_gpu=True)
        self._env_id = env_id
        self._num_envs = int(num_envs)
        self._device = torch.device(device)
        self._obs_dim, self._act_dim = SYNTH_DIMS[env_id]
        self._horizon, self._cost_p = int(horizon), float(cost_p)
        self._observation_space = Box(-np.inf, np.inf, (self._obs_dim,))
        self._action_space = Box(-1.0, 1.0, (self._act_dim,))
        self._seed = int(seed)
        # Philox stream position = *_t_base (device) + _t (host, passed by value).  commit() folds the host part
        # into the device part: a captured hipGraph of an epoch replays with the same by-value positions
        # 0..T while the device part advances, so every epoch still draws fresh numbers
        self._t = 0
        self._t_base = torch.zeros(1, dtype=torch.int64, device=torch.device(device))
        self._since_reset = 0  # all envs reset together -> truncation steps are known on the host
        N, dev = self._num_envs, self._device
        f32 = dict(dtype=torch.float32, device=dev)
        self._steps = torch.zeros(N, dtype=torch.int32, device=dev)
        self._obs = [torch.empty(N, self._obs_dim, **f32) for _ in range(2)]
        self._final = torch.zeros(N, self._obs_dim, **f32)
        self._reward, self._cost = torch.empty(N, **f32), torch.empty(N, **f32)
        self._term = torch.zeros(N, dtype=torch.uint8, device=dev)
        self._trunc = torch.zeros(N, dtype=torch.uint8, device=dev)
        self._flip = 0

    num_envs = property(lambda self: self._num_envs)
    observation_space = property(lambda self: self._observation_space)
    action_space = property(lambda self: self._action_space)
    max_episode_steps = property(lambda self: self._horizon)

    def set_seed(self, seed: int) -> None:
        self._seed = int(seed)

    def _launch(self, obs, reset_only: int) -> None:
        _lib.check(self._lib.osa_synth_env_step(
            self._seed & 0xFFFFFFFFFFFFFFFF, self._t, _lib.ptr(self._t_base), self._num_envs, self._obs_dim,
            self._horizon, self._cost_p, _lib.ptr(self._steps), _lib.ptr(obs), self._obs_dim, _lib.ptr(self._reward),
            _lib.ptr(self._cost), _lib.ptr(self._term), _lib.ptr(self._trunc), _lib.ptr(self._final),
            self._obs_dim, reset_only, _lib.stream_ptr()), 'osa_synth_env_step')
        self._t += 1

    def reset(self, seed: int | None = None, options: dict | None = None):
        if seed is not None:
            self.set_seed(seed)
        self._flip ^= 1
        obs = self._obs[self._flip]
        self._launch(obs, 1)
        self._since_reset = 0
        return obs, {}

    def step(self, action: torch.Tensor):
        self._flip ^= 1
        obs = self._obs[self._flip]
        self._launch(obs, 0)
        self._since_reset += 1
        info: dict[str, Any] = {}
        if self._since_reset % self._horizon == 0:  # every env truncates on this step
            info['final_observation'] = self._final
            info['_final_observation'] = self._trunc
        return obs, self._reward, self._cost, self._term, self._trunc, info


    graph_safe = True  # every step is a fixed sequence of launches on device tensors (no host-side data flow)

    def commit(self) -> None:
        """Fold the host part of the stream position into the device part (end of an epoch; capturable)."""
        self._t_base += self._t
        self._t = 0

    def render(self):
        return None

    def close(self) -> None:
        return None


@env_register
class ReachVectorEnv:  # pylint: disable=too-many-instance-attributes
    """Learnable synthetic vector CMDP in HBM (osa_reach_env_step): a point reaches resampled goals
    (reward = progress, +1 per goal) past a hazard disc (cost 1 inside).  Observation/action dims of
    SafetyPointGoal1 (60 / 2).  Used for learning-curve comparisons against the reference, which trains
    on the CPU twin of this env kept with the test harness under oracle/."""

    _support_envs = ['SynthReach-v0']
    need_auto_reset_wrapper = False
    need_time_limit_wrapper = False
    need_evaluation = False

    def __init__(self, env_id: str, num_envs: int = 1, device='cuda:0', horizon: int = 50,
                 seed: int = 0, **_unused) -> None:
        self._lib = _lib.load(require_gpu=True)
        self._num_envs = int(num_envs)
        self._device = torch.device(device)
        self._obs_dim, self._act_dim = 60, 2
        self._horizon = int(horizon)
        self._observation_space = Box(-np.inf, np.inf, (self._obs_dim,))
        self._action_space = Box(-1.0, 1.0, (self._act_dim,))
        self._seed = int(seed)
        self._t = 0            # host part of the Philox stream position (see SynthVectorEnv)
        self._t_base = torch.zeros(1, dtype=torch.int64, device=self._device)
        self._since_reset = 0
        N, dev = self._num_envs, self._device
        f32 = dict(dtype=torch.float32, device=dev)
        self.state = torch.zeros(N, 8, **f32)  # p, goal, hazard, pad
        self._steps = torch.zeros(N, dtype=torch.int32, device=dev)
        self._obs = [torch.empty(N, self._obs_dim, **f32) for _ in range(2)]
        self._final = torch.zeros(N, self._obs_dim, **f32)
        self._reward, self._cost = torch.empty(N, **f32), torch.empty(N, **f32)
        self._term = torch.zeros(N, dtype=torch.uint8, device=dev)
        self._trunc = torch.zeros(N, dtype=torch.uint8, device=dev)
        self._flip = 0

    num_envs = property(lambda self: self._num_envs)
    observation_space = property(lambda self: self._observation_space)
    action_space = property(lambda self: self._action_space)
    max_episode_steps = property(lambda self: self._horizon)

    def set_seed(self, seed: int) -> None:
        self._seed = int(seed)

    def _launch(self, obs, action, reset_only: int) -> None:
        ld_a = action.stride(0) if action is not None else 0
        _lib.check(self._lib.osa_reach_env_step(
            self._seed & 0xFFFFFFFFFFFFFFFF, self._t, _lib.ptr(self._t_base), self._num_envs, self._obs_dim,
            self._horizon, _lib.ptr(self.state), _lib.ptr(self._steps), _lib.ptr(action), ld_a, _lib.ptr(obs),
            self._obs_dim, _lib.ptr(self._reward), _lib.ptr(self._cost), _lib.ptr(self._term),
            _lib.ptr(self._trunc), _lib.ptr(self._final), self._obs_dim, reset_only,
            _lib.stream_ptr()), 'osa_reach_env_step')
        self._t += 1

    def reset(self, seed: int | None = None, options: dict | None = None):
        if seed is not None:
            self.set_seed(seed)
        self._flip ^= 1
        obs = self._obs[self._flip]
        self._launch(obs, None, 1)
        self._since_reset = 0
        return obs, {}

    def step(self, action: torch.Tensor):
        assert action.dtype == torch.float32 and action.shape == (self._num_envs, self._act_dim) \
            and action.stride(1) == 1
        self._flip ^= 1
        obs = self._obs[self._flip]
        self._launch(obs, action, 0)
        self._since_reset += 1
        info: dict[str, Any] = {}
        if self._since_reset % self._horizon == 0:  # every env truncates on this step
            info['final_observation'] = self._final
            info['_final_observation'] = self._trunc
        return obs, self._reward, self._cost, self._term, self._trunc, info


    graph_safe = True  # every step is a fixed sequence of launches on device tensors (no host-side data flow)

    def commit(self) -> None:
        """Fold the host part of the stream position into the device part (end of an epoch; capturable)."""
        self._t_base += self._t
        self._t = 0

    def render(self):
        return None

    def close(self) -> None:
        return None
