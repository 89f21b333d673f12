"""TEST INFRASTRUCTURE ONLY -- lets the *unmodified* reference (PKU-Alignment/omnisafe) execute: in the build
container from /root/reference (to pin the oracle and generate golden vectors), on the GPU box -- which has no
/root/reference -- from the archive oracle/_ref/omnisafe_ref.zip that oracle/stage_reference.py writes at build
time (git-ignored build output; unpacked into a scratch directory here).

Nothing in the product (``omnisafe_amd/``) imports this file.  Users: ``oracle/make_golden.py`` (fixture
generator), the ``-m "not gpu"`` tests that cross-check ``oracle/np_oracle.py`` against the live reference,
``tests/test_reference_facade_gpu.py`` (the reference's own ``omnisafe.Agent`` driving the plugin on the GPU) and
``oracle/ref_cpu_baseline.py`` (bench.py's ``cpu_baseline``, kind "reference").  Every user skips (or reports the
baseline as unavailable) when neither source of the reference exists.

What it does:
  1. appends a meta-path finder that fabricates empty stand-in modules for the reference's third-party
     imports that are not installed here (gymnasium, safety_gymnasium, wandb, tensorboard, ...).  Real
     packages win when present because the finder is *appended*.
  2. pre-seeds ``torch.utils.tensorboard`` with a no-op ``SummaryWriter``
     (reference omnisafe/common/logger.py:49 imports it unconditionally).
  3. puts /root/reference on sys.path.
  4. ``register_synth_env()`` registers a synthetic vector CMDP (zero-cost stand-in for
     Safety-Gymnasium, same distributions as omnisafe_amd's device env: obs ~ N(0,1)^D_o,
     reward ~ N(0,1), cost ~ Bernoulli(p), truncation every ``horizon`` steps).
  5. ``register_reach_env()`` registers the learnable point-reach CMDP ``SynthReach-v0`` (dynamics:
     oracle/np_oracle.py reach_env_step) used for the learning-curve comparison.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import os
import sys
import types

_DEFAULT_ROOT = os.environ.get('OMNISAFE_REFERENCE_ROOT', '/root/reference')
_HERE = os.path.dirname(os.path.abspath(__file__))
STAGED_ARCHIVE = os.path.join(_HERE, '_ref', 'omnisafe_ref.zip')  # written by oracle/stage_reference.py
REFERENCE_ROOT = _DEFAULT_ROOT

_STUB_PREFIXES = (
    'gymnasium', 'safety_gymnasium', 'wandb', 'tensorboard', 'pytorch_lightning', 'moviepy',
    'seaborn', 'gdown', 'cvxopt', 'gpytorch', 'qpth', 'isaacgym', 'metadrive', 'matplotlib',
    'gymnasium_robotics', 'mujoco', 'tensorboardX',
)


def _unpack_staged() -> str | None:
    """GPU box: /root/reference does not exist there; the build container staged the unmodified package
    as oracle/_ref/omnisafe_ref.zip (git-ignored build output).  The reference reads its YAML files
    through the file system (omnisafe/utils/config.py:250-262), so the archive is unpacked into a scratch
    directory outside the repository instead of being imported in place."""
    if not os.path.exists(STAGED_ARCHIVE):
        return None
    cached = os.environ.get('OSA_REF_UNPACKED')
    if cached and os.path.isdir(os.path.join(cached, 'omnisafe')):
        return cached
    import tempfile
    import zipfile

    root = tempfile.mkdtemp(prefix='osa_reference_')
    with zipfile.ZipFile(STAGED_ARCHIVE) as z:
        z.extractall(root)
    os.environ['OSA_REF_UNPACKED'] = root  # child processes (torchrun ranks, subprocess benches) reuse it
    return root


def reference_root() -> str | None:
    """Directory that holds the unmodified `omnisafe/` package, or None."""
    global REFERENCE_ROOT
    if os.path.isdir(os.path.join(_DEFAULT_ROOT, 'omnisafe')):
        REFERENCE_ROOT = _DEFAULT_ROOT
        return REFERENCE_ROOT
    root = _unpack_staged()
    if root:
        REFERENCE_ROOT = root
    return root


def reference_available() -> bool:
    return reference_root() is not None


class _Box:
    """Tiny stand-in for gymnasium.spaces.Box (only .low/.high/.shape/.dtype/.sample are used on the
    hot path: reference omnisafe/common/buffer/base.py:73-80, omnisafe/envs/wrapper.py:451-480)."""

    def __init__(self, low, high, shape=None, dtype=None, seed=None):
        import numpy as np

        if shape is None:
            shape = np.asarray(low).shape
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype) if dtype is not None else np.dtype('float32')
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

    def sample(self):
        import numpy as np

        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return np.random.uniform(lo, hi).astype(self.dtype)


class _Discrete:
    def __init__(self, n, seed=None, start=0):
        self.n = int(n)
        self.shape = ()

    def sample(self):
        import numpy as np

        return int(np.random.randint(self.n))


class _StubModule(types.ModuleType):
    __path__: list = []

    def __getattr__(self, name):
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        if name == 'Box':
            value = _Box
        elif name == 'Discrete':
            value = _Discrete
        else:
            value = type(name, (object,), {'__init__': lambda self, *a, **k: None})
        setattr(self, name, value)
        return value


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split('.')[0] in _STUB_PREFIXES:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        mod = _StubModule(spec.name)
        mod.__path__ = []
        return mod

    def exec_module(self, module):
        return None


_installed = False


def install() -> None:
    """Make ``import omnisafe`` resolve to the unmodified reference."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(f'reference not found at {_DEFAULT_ROOT} and no staged archive at {STAGED_ARCHIVE}')
    sys.meta_path.append(_StubFinder())
    tb = types.ModuleType('torch.utils.tensorboard')
    tbw = types.ModuleType('torch.utils.tensorboard.writer')

    class SummaryWriter:  # no-op
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

        def flush(self):
            pass

        def close(self):
            pass

    tb.SummaryWriter = SummaryWriter
    tbw.SummaryWriter = SummaryWriter
    tb.writer = tbw
    sys.modules.setdefault('torch.utils.tensorboard', tb)
    sys.modules.setdefault('torch.utils.tensorboard.writer', tbw)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import gymnasium.spaces  # noqa: F401  (resolve the stub *submodule* first, Appendix B item 4)

    _installed = True


def import_reference():
    install()
    import omnisafe  # noqa: F401

    return omnisafe


def import_simple_env():
    """The reference's own test env (tests/simple_env.py:30-90, id 'Test-v0': single env that asks for the
    TimeLimit and AutoReset wrappers), registered with the reference's env registry by its `@env_register`."""
    import importlib.util

    import_reference()
    if 'ref_simple_env' in sys.modules:
        return sys.modules['ref_simple_env']
    path = os.path.join(reference_root(), 'tests', 'simple_env.py')
    spec = importlib.util.spec_from_file_location('ref_simple_env', path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules['ref_simple_env'] = mod
    spec.loader.exec_module(mod)
    return mod


_SYNTH_REGISTERED = False
DEFAULT_HORIZON = 1000  # used when env_cfgs carries no 'horizon' (TRPOLag/CPO YAMLs have no env_cfgs key)


def register_synth_env():
    """Register ``Synth{D_o}x{D_a}-v0`` style ids with the reference's env registry.

    Env id grammar: ``SynthPointGoal1-v0`` (60/2), ``SynthCarGoal1-v0`` (72/2), ``SynthAnt-v0`` (27/8),
    ``SynthHumanoid-v0`` (376/17), ``SynthTiny-v0`` (6/2).  ``env_cfgs`` understood: horizon, cost_p.
    """
    global _SYNTH_REGISTERED
    install()
    import torch
    from omnisafe.envs.core import CMDP, env_register
    from gymnasium.spaces import Box

    if _SYNTH_REGISTERED:
        return
    dims = {
        'SynthPointGoal1-v0': (60, 2),
        'SynthCarGoal1-v0': (72, 2),
        'SynthAnt-v0': (27, 8),
        'SynthHumanoid-v0': (376, 17),
        'SynthTiny-v0': (6, 2),
    }

    @env_register
    class SynthRefEnv(CMDP):  # pylint: disable=too-many-instance-attributes
        _support_envs = list(dims)
        need_auto_reset_wrapper = False
        need_time_limit_wrapper = False
        need_evaluation = False

        def __init__(self, env_id, num_envs=1, device=torch.device('cpu'), **kwargs):
            super().__init__(env_id)
            self._num_envs = num_envs
            self._device = torch.device(device)
            d_o, d_a = dims[env_id]
            self._d_o, self._d_a = d_o, d_a
            self._horizon = int(kwargs.get('horizon', DEFAULT_HORIZON))
            self._cost_p = float(kwargs.get('cost_p', 0.05))
            self._observation_space = Box(-float('inf'), float('inf'), (d_o,))
            self._action_space = Box(-1.0, 1.0, (d_a,))
            self._metadata = {}
            self._gen = torch.Generator(device='cpu')
            self._gen.manual_seed(0)
            self._steps = torch.zeros(num_envs, dtype=torch.int64)

        @property
        def max_episode_steps(self):
            return self._horizon

        def _obs(self):
            return torch.randn(self._num_envs, self._d_o, generator=self._gen).to(self._device)

        def set_seed(self, seed):
            self._gen.manual_seed(int(seed))

        def reset(self, seed=None, options=None):
            if seed is not None:
                self.set_seed(seed)
            self._steps.zero_()
            obs = self._obs()
            if self._num_envs == 1:
                obs = obs[0]
            return obs, {}

        def step(self, action):
            n = self._num_envs
            obs = self._obs()
            reward = torch.randn(n, generator=self._gen).to(self._device)
            cost = (torch.rand(n, generator=self._gen) < self._cost_p).float().to(self._device)
            self._steps += 1
            truncated = self._steps >= self._horizon
            terminated = torch.zeros(n, dtype=torch.bool)
            info = {}
            if bool(truncated.any()):
                info['final_observation'] = obs.clone()
                info['_final_observation'] = truncated.clone()
                fresh = self._obs()
                obs = torch.where(truncated[:, None].to(self._device), fresh, obs)
                self._steps[truncated] = 0
            if n == 1:
                return (obs[0], reward[0], cost[0], terminated[0].to(self._device),
                        truncated[0].to(self._device),
                        {k: (v[0] if k == 'final_observation' else v) for k, v in info.items()})
            return obs, reward, cost, terminated.to(self._device), truncated.to(self._device), info

        def render(self):
            return None

        def close(self):
            return None

    _SYNTH_REGISTERED = True
    return SynthRefEnv


_REACH_REGISTERED = False
_REACH_CLS = None


def register_reach_env():
    """Register ``SynthReach-v0`` (obs 60 / act 2) with the reference's env registry: the CPU twin of
    omnisafe_amd's device env of the same id.  Dynamics from oracle/np_oracle.py; resets and goal
    resampling draw from a numpy Generator seeded by ``set_seed``."""
    global _REACH_REGISTERED, _REACH_CLS
    install()
    import numpy as np
    import torch
    from omnisafe.envs.core import CMDP, env_register
    from gymnasium.spaces import Box

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import np_oracle

    if _REACH_REGISTERED:
        return _REACH_CLS

    @env_register
    class ReachRefEnv(CMDP):  # pylint: disable=too-many-instance-attributes
        _support_envs = ['SynthReach-v0']
        need_auto_reset_wrapper = False
        need_time_limit_wrapper = False
        need_evaluation = False

        def __init__(self, env_id, num_envs=1, device=torch.device('cpu'), **kwargs):
            super().__init__(env_id)
            self._num_envs = num_envs
            self._device = torch.device(device)
            self._d_o = 60
            self._horizon = int(kwargs.get('horizon', 50))
            self._observation_space = Box(-float('inf'), float('inf'), (self._d_o,))
            self._action_space = Box(-1.0, 1.0, (2,))
            self._metadata = {}
            self._rng = np.random.default_rng(0)
            self._state = np.zeros((num_envs, 6), np.float32)
            self._steps = 0

        @property
        def max_episode_steps(self):
            return self._horizon

        def set_seed(self, seed):
            self._rng = np.random.default_rng(int(seed))

        def _draw(self, n, k):
            return self._rng.uniform(-1.0, 1.0, size=(n, k)).astype(np.float32)

        def _obs(self):
            return torch.from_numpy(np_oracle.reach_env_obs(self._state, self._d_o)).to(self._device)

        def reset(self, seed=None, options=None):
            if seed is not None:
                self.set_seed(seed)
            self._state = self._draw(self._num_envs, 6)
            self._steps = 0
            obs = self._obs()
            return (obs[0] if self._num_envs == 1 else obs), {}

        def step(self, action):
            n = self._num_envs
            act = action.detach().cpu().numpy().reshape(n, -1)
            q, reward, cost, reached = np_oracle.reach_env_step(self._state, act)
            self._state[:, 0:2] = q
            if reached.any():
                self._state[reached, 2:4] = self._draw(int(reached.sum()), 2)
            self._steps += 1
            obs = self._obs()
            done = self._steps >= self._horizon
            truncated = torch.full((n,), bool(done))
            terminated = torch.zeros(n, dtype=torch.bool)
            info = {}
            if done:
                info['final_observation'] = obs.clone()
                info['_final_observation'] = truncated.clone()
                self._state = self._draw(n, 6)
                self._steps = 0
                obs = self._obs()
            reward_t = torch.from_numpy(reward).to(self._device)
            cost_t = torch.from_numpy(cost).to(self._device)
            if n == 1:
                info = {k: (v[0] if k == 'final_observation' else v) for k, v in info.items()}
                return obs[0], reward_t[0], cost_t[0], terminated[0], truncated[0], info
            return obs, reward_t, cost_t, terminated.to(self._device), truncated.to(self._device), info

        def render(self):
            return None

        def close(self):
            return None

    _REACH_REGISTERED = True
    _REACH_CLS = ReachRefEnv
    return ReachRefEnv
