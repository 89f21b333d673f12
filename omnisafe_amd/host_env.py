"""Host (CPU) environments behind the device-resident adapter.

The north star puts vectorised Safety-Gymnasium envs -- MuJoCo on the host -- behind
``OnPolicyAdapter.rollout``, and the reference lets users register any ``CMDP`` of their own with
``@env_register`` (omnisafe/envs/core.py:300-421, template envs/custom_env.py, test double
tests/simple_env.py:30-90).  Such an env lives on the host; everything else of the step (policy, wrappers'
arithmetic, buffer, bootstrap values, episode accounting) stays on the device.  :class:`HostEnvBridge` is the
boundary the reference crosses at envs/safety_gymnasium_env.py:190-210 (``action.detach().cpu().numpy()`` down,
five ``torch.as_tensor(..., device=...)`` up), restated as ONE device-to-host copy and ONE host-to-device copy
per vector step through pinned staging buffers:

    down  action (N, D_a) float32                                  4 D_a           bytes per env-step
    up    obs (N, D_o) | reward | cost | terminated | truncated | final mask   4 (D_o + 5)    bytes per env-step
          + final_observation (N, D_o), only on steps where some env finished  (+ 4 D_o)

The host-side wrappers the reference applies to single envs that ask for them -- ``TimeLimit`` then
``AutoReset`` (adapter/online_adapter.py:120-132, envs/wrapper.py:31-176) -- wrap the HOST env, below the
bridge: their per-step ``bool(terminated) or bool(truncated)`` tests read host memory and never synchronise
the device.
"""
from __future__ import annotations

import numpy as np
import torch


class _EnvWrapper:
    """omnisafe/envs/core.py:185-297 (Wrapper): forwards everything it does not override."""

    graph_safe = False  # host-side decisions per step

    def __init__(self, env, device) -> None:
        self._env, self._device = env, torch.device(device)

    def __getattr__(self, name):
        return getattr(self._env, name)

    def reset(self, seed=None, options=None):
        return self._env.reset(seed=seed, options=options)

    def step(self, action):
        return self._env.step(action)


class TimeLimit(_EnvWrapper):
    """omnisafe/envs/wrapper.py:31-107: truncated = (steps since reset >= time_limit); single env."""

    need_time_limit_wrapper = False

    def __init__(self, env, time_limit: int, device) -> None:
        super().__init__(env, device)
        assert int(env.num_envs) == 1, 'TimeLimit only supports single environment'
        self._time, self._time_limit = 0, int(time_limit)

    def reset(self, seed=None, options=None):
        self._time = 0
        return self._env.reset(seed=seed, options=options)

    def step(self, action):
        obs, reward, cost, terminated, truncated, info = self._env.step(action)
        self._time += 1
        truncated = torch.tensor(self._time >= self._time_limit, dtype=torch.bool, device=self._device)
        return obs, reward, cost, terminated, truncated, info


class AutoReset(_EnvWrapper):
    """omnisafe/envs/wrapper.py:110-176: on terminated / truncated the env is reset, the returned observation is
    the first of the new episode and the true last one goes to info['final_observation']; single env (one host
    read of the two flags per step, as in the reference)."""

    need_auto_reset_wrapper = False

    def __init__(self, env, device) -> None:
        super().__init__(env, device)
        assert int(env.num_envs) == 1, 'AutoReset only supports single environment'

    def step(self, action):
        obs, reward, cost, terminated, truncated, info = self._env.step(action)
        if bool(torch.as_tensor(terminated).any()) or bool(torch.as_tensor(truncated).any()):
            new_obs, new_info = self._env.reset()
            assert 'final_observation' not in new_info, 'info dict cannot contain key "final_observation" '
            assert 'final_info' not in new_info, 'info dict cannot contain key "final_info" '
            new_info = dict(new_info)
            new_info['final_observation'] = obs
            new_info['final_info'] = info
            obs, info = new_obs, new_info
        return obs, reward, cost, terminated, truncated, info


def _host_f32(x, shape) -> torch.Tensor:
    """Whatever a host env returns for one field (tensor, numpy array, Python scalar) as a CPU float32 tensor."""
    if isinstance(x, torch.Tensor):
        t = x.detach()
        if t.device.type != 'cpu':
            t = t.cpu()
    else:
        t = torch.as_tensor(np.asarray(x))
    return t.to(torch.float32).reshape(shape)


class HostEnvBridge:  # pylint: disable=too-many-instance-attributes
    """Presents a host ``CMDP`` (reference interface, envs/core.py:37-182) as the device env the adapter drives.

    ``step(action)`` takes the DEVICE action tensor (N, D_a) and returns DEVICE tensors with the vector
    auto-reset convention of :mod:`omnisafe_amd.envs` (``info['final_observation']`` (N, D_o) +
    ``info['_final_observation']`` (N,) on steps where some env finished).  A single env (``num_envs == 1``)
    is stepped with the squeezed action and its scalar outputs are unsqueezed, as the reference's
    ``Unsqueeze`` wrapper does (envs/wrapper.py:568-637); whenever it reports a ``final_observation`` the whole
    (one-row) batch is marked, which is the reference's ``slice(None)`` (envs/wrapper.py:233).
    """

    graph_safe = False
    need_auto_reset_wrapper = False
    need_time_limit_wrapper = False
    host_resident = True

    def __init__(self, env, device) -> None:
        self._device = torch.device(device)
        cpu = torch.device('cpu')
        # online_adapter.py:120-132, on the host side of the boundary
        if getattr(env, 'need_time_limit_wrapper', False):
            assert env.max_episode_steps, ('You must define max_episode_steps as an integer\n'
                                           'or cancel the use of the time_limit wrapper.')
            env = TimeLimit(env, time_limit=int(env.max_episode_steps), device=cpu)
        if getattr(env, 'need_auto_reset_wrapper', False):
            env = AutoReset(env, device=cpu)
        self._env = env
        self._num_envs = N = int(env.num_envs)
        obs_space, act_space = env.observation_space, env.action_space
        if len(obs_space.shape) != 1 or len(act_space.shape) != 1:
            raise NotImplementedError('only flat Box observation / action spaces')  # buffer/base.py:73-80
        self._obs_dim = Do = int(obs_space.shape[0])
        self._act_dim = Da = int(act_space.shape[0])
        pin = self._device.type == 'cuda'
        # ---- staging, one flat block so that a step is ONE host-to-device copy:
        #   [ obs N*Do | reward N | cost N | terminated N | truncated N | final mask N | final_obs N*Do ]
        # the final block is the tail: on steps where no env finished only the head travels
        self._head = N * Do + 5 * N
        total = self._head + N * Do
        self._up_h = torch.zeros(total, dtype=torch.float32, pin_memory=pin)
        self._up_d = torch.zeros(total, dtype=torch.float32, device=self._device)
        self._act_h = torch.zeros(N, Da, dtype=torch.float32, pin_memory=pin)
        o = 0
        self._v: dict[str, tuple[torch.Tensor, torch.Tensor]] = {}
        for name, n, shape in (('obs', N * Do, (N, Do)), ('reward', N, (N,)), ('cost', N, (N,)),
                               ('terminated', N, (N,)), ('truncated', N, (N,)), ('fmask', N, (N,)),
                               ('final', N * Do, (N, Do))):
            self._v[name] = (self._up_h[o:o + n].view(shape), self._up_d[o:o + n].view(shape))
            o += n
        self.bytes_down = 0  # running totals (PCIe bytes per env-step); reset by the caller at will
        self.bytes_up = 0
        self.steps = 0
        # device work of the PREVIOUS vector step that the next action does not depend on (bootstrap values, episode
        # accounting: adapter.py): enqueued behind the action's device-to-host copy, so that it runs while the host
        # steps the env instead of in front of the next policy step
        self.deferred_device_work = None
        self._copied = torch.cuda.Event() if pin else None
        # optional wall-clock split of a step (tools/host_env_bridge_timing.py): seconds accumulated per phase
        self.timing: dict | None = None

    # ------------------------------------------------------------------ reference surface
    def __getattr__(self, name):  # spaces, env_spec_log, max_episode_steps, render, need_evaluation, ...
        if name.startswith('_'):
            raise AttributeError(name)
        return getattr(self._env, name)

    @property
    def num_envs(self) -> int:
        return self._num_envs

    @property
    def host_env(self):
        return self._env

    def defer(self, work) -> None:
        """Hand over device work of the step just taken that the next action does not depend on; it is enqueued behind
        the next action's device-to-host copy (step) or in front of a reset.  A method, so that env wrappers which
        forward attribute READS to the bridge (adapter._EarlyTerminatedEnv) reach it."""
        if self.deferred_device_work is not None:  # (never dropped: run what is still pending first, in order)
            pending, self.deferred_device_work = self.deferred_device_work, None
            pending()
        self.deferred_device_work = work

    def set_seed(self, seed: int) -> None:
        self._env.set_seed(seed)

    def close(self) -> None:
        self._env.close()

    def _upload(self, n_floats: int) -> None:
        self._up_d[:n_floats].copy_(self._up_h[:n_floats], non_blocking=True)
        self.bytes_up += 4 * n_floats

    def reset(self, seed: int | None = None, options: dict | None = None):
        if self.deferred_device_work is not None:
            work, self.deferred_device_work = self.deferred_device_work, None
            work()
        obs, info = self._env.reset(seed=seed, options=options)
        if self._device.type == 'cuda':
            torch.cuda.current_stream(self._device).synchronize()  # an earlier upload may still read the staging block
        h, d = self._v['obs']
        h.copy_(_host_f32(obs, h.shape))
        self._upload(h.numel())
        return d, dict(info)

    def step(self, action: torch.Tensor):
        N = self._num_envs
        # ---- down: one device-to-host copy; its completion is the one host synchronisation of the step (it also
        # orders this step's staging writes after the previous step's upload, which sits earlier in the stream)
        import time

        tm = self.timing
        t0 = time.perf_counter() if tm is not None else 0.0
        self._act_h.copy_(action.detach().reshape(N, self._act_dim), non_blocking=True)
        if self._device.type == 'cuda':
            # wait for the COPY only (an event behind it), not for the stream: what the previous step left to do on the
            # device goes out first and overlaps the host env's step
            self._copied.record(torch.cuda.current_stream(self._device))
            work, self.deferred_device_work = self.deferred_device_work, None
            if work is not None:
                work()
            self._copied.synchronize()
        elif self.deferred_device_work is not None:
            work, self.deferred_device_work = self.deferred_device_work, None
            work()
        t1 = time.perf_counter() if tm is not None else 0.0
        self.bytes_down += 4 * self._act_h.numel()
        act = self._act_h[0] if N == 1 else self._act_h  # Unsqueeze.step squeezes the action (wrapper.py:600)
        obs, reward, cost, terminated, truncated, info = self._env.step(act)
        t2 = time.perf_counter() if tm is not None else 0.0
        v = self._v
        v['obs'][0].copy_(_host_f32(obs, (N, self._obs_dim)))
        v['reward'][0].copy_(_host_f32(reward, (N,)))
        v['cost'][0].copy_(_host_f32(cost, (N,)))
        v['terminated'][0].copy_(_host_f32(terminated, (N,)))
        v['truncated'][0].copy_(_host_f32(truncated, (N,)))
        out_info: dict = {k: val for k, val in info.items()
                          if k not in ('final_observation', '_final_observation')}
        have_final = 'final_observation' in info
        if have_final:
            fo = info['final_observation']
            if not isinstance(fo, torch.Tensor):  # gymnasium's object array with None for unfinished envs
                fo = np.array([np.zeros(self._obs_dim, np.float32) if a is None else np.asarray(a, np.float32)
                               for a in (fo if N > 1 else [fo])])  # safety_gymnasium_env.py:197-208
            v['final'][0].copy_(_host_f32(fo, (N, self._obs_dim)))
            if N > 1 and '_final_observation' in info:
                v['fmask'][0].copy_(_host_f32(info['_final_observation'], (N,)))
            else:
                v['fmask'][0].fill_(1.0)  # single env: the reference normalises the whole (one-row) slice
            self._upload(self._up_h.numel())
        else:
            v['fmask'][0].zero_()
            self._upload(self._head)
        self.steps += 1
        if tm is not None:
            t3 = time.perf_counter()
            tm['wait_device_and_d2h'] = tm.get('wait_device_and_d2h', 0.0) + (t1 - t0)
            tm['host_env_step'] = tm.get('host_env_step', 0.0) + (t2 - t1)
            tm['staging_and_h2d_enqueue'] = tm.get('staging_and_h2d_enqueue', 0.0) + (t3 - t2)
        if have_final:
            out_info['final_observation'] = v['final'][1]
            out_info['_final_observation'] = v['fmask'][1]
        return (v['obs'][1], v['reward'][1], v['cost'][1], v['terminated'][1], v['truncated'][1], out_info)

    def pcie_bytes_per_env_step(self) -> tuple[float, float]:
        """(down, up) bytes per env-step averaged over the steps taken so far."""
        n = max(self.steps, 1) * self._num_envs
        return self.bytes_down / n, self.bytes_up / n


def make_reference_env(env_id: str, num_envs: int, device, **env_cfgs):
    """envs/core.py:389-421 for ids this package does not own: build the env through the CALLER's ``omnisafe``
    (the reference's registry, where Safety-Gymnasium and every ``@env_register`` class of the user live) on the
    host, and put the bridge on top.  Returns None when the reference is not importable or does not know the id.
    The reference package is looked up, never shipped: nothing here depends on it when it is absent."""
    import sys

    core = sys.modules.get('omnisafe.envs.core')
    if core is None:
        try:
            import omnisafe.envs.core as core  # noqa: PLC0415  the caller's installation
        except ModuleNotFoundError as exc:
            if exc.name is not None and exc.name.split('.')[0] == 'omnisafe':
                return None  # the reference is not installed: no reference-registered envs
            raise ImportError(f'omnisafe is installed but failed to import while looking up {env_id!r} (missing '
                              f'dependency {exc.name!r})') from exc
        # (any other exception -- a broken reference installation -- propagates as what it is instead of being
        # reported as "env not registered": round-3 advisor finding)
    if env_id not in core.support_envs():
        return None
    host = core.make(env_id, num_envs=num_envs, device=torch.device('cpu'), **env_cfgs)
    return HostEnvBridge(host, device)
