"""CPU tests of host-side logic that involves no device arithmetic: config merge, registry, Lagrange
scalar algebra (vs the reference), logger csv columns, plugin installation behind the reference's
omnisafe.Agent, no-GPU failure mode."""
import csv
import os
import sys

import numpy as np
import pytest
import torch

import np_oracle as O


def test_config_defaults_match_reference_yaml_values():
    from omnisafe_amd.config import get_default_kwargs

    d = get_default_kwargs('PPOLag')
    a = d['algo_cfgs']
    assert (a['steps_per_epoch'], a['update_iters'], a['batch_size']) == (20000, 40, 64)
    assert (a['target_kl'], a['clip'], a['gamma'], a['lam'], a['lam_c']) == (0.02, 0.2, 0.99, 0.95, 0.95)
    assert d['lagrange_cfgs'] == {'cost_limit': 25.0, 'lagrangian_multiplier_init': 0.001,
                                  'lambda_lr': 0.035, 'lambda_optimizer': 'Adam'}
    t = get_default_kwargs('TRPOLag')
    assert t['model_cfgs']['actor']['lr'] is None and t['model_cfgs']['critic']['lr'] == 0.001
    assert (t['algo_cfgs']['cg_iters'], t['algo_cfgs']['cg_damping'], t['algo_cfgs']['batch_size']) == (15, 0.1, 128)
    c = get_default_kwargs('CPO')
    assert c['algo_cfgs']['cost_limit'] == 25.0 and 'lagrange_cfgs' not in c


@pytest.mark.reference
def test_config_defaults_equal_reference_yaml():
    """Every default of the three accelerated algorithms equals the reference's YAML ``defaults``."""
    import yaml

    from omnisafe_amd.config import get_default_kwargs

    for algo in ('PPOLag', 'TRPOLag', 'CPO'):
        ref = yaml.safe_load(open(f'/root/reference/omnisafe/configs/on-policy/{algo}.yaml'))['defaults']
        mine = get_default_kwargs(algo)

        def walk(r, m, path):
            for k, v in r.items():
                if path + [k] in (['train_cfgs', 'device'], ['logger_cfgs', 'use_tensorboard']):
                    continue  # deliberate: GPU device, no tensorboard dependency
                assert k in m, path + [k]
                if isinstance(v, dict):
                    walk(v, m[k], path + [k])
                else:
                    assert m[k] == v, (algo, path + [k], m[k], v)

        walk(ref, mine, [])


def test_custom_cfg_validation():
    import omnisafe_amd
    from omnisafe_amd.config import get_default_kwargs, recursive_check_config

    recursive_check_config({'algo_cfgs': {'batch_size': 128}}, get_default_kwargs('PPOLag'))
    with pytest.raises(KeyError):
        recursive_check_config({'algo_cfgs': {'bogus': 1}}, get_default_kwargs('PPOLag'))
    with pytest.raises(AssertionError):
        omnisafe_amd.Agent('NoSuchAlgo', 'SynthTiny-v0')


def test_registry_semantics():
    from omnisafe_amd.algorithms import registry

    assert registry.get('PPOLag').__name__ == 'PPOLag'
    with pytest.raises(KeyError):
        registry.REGISTRY._register_module(registry.get('PPOLag'))  # duplicate name
    with pytest.raises(KeyError):
        registry.get('Nope')


def test_lagrange_matches_oracle_trajectory():
    from omnisafe_amd.lagrange import Lagrange

    rng = np.random.default_rng(0)
    mine = Lagrange(25.0, 0.001, 0.035)
    ref = O.Lagrange(25.0, 0.001, 0.035)
    for Jc in rng.uniform(0, 60, size=50):
        mine.update_lagrange_multiplier(float(Jc))
        ref.update_lagrange_multiplier(float(Jc))
        assert np.float32(mine.lagrangian_multiplier) == np.float32(ref.lagrangian_multiplier.item())
    up = Lagrange(25.0, 0.5, 0.1, lagrangian_upper_bound=0.6)
    for _ in range(20):
        up.update_lagrange_multiplier(100.0)
    assert up.lagrangian_multiplier == pytest.approx(0.6)


def test_lagrange_vs_reference_golden(golden):
    from omnisafe_amd.lagrange import Lagrange

    g = golden('ppolag_epoch.npz')
    lag = Lagrange(25.0, 0.001, 0.035)
    assert np.float32(lag.lagrangian_multiplier) == g['update/lambda_before']
    lag.update_lagrange_multiplier(float(g['update/Jc']))
    assert np.float32(lag.lagrangian_multiplier) == g['update/lambda_after']


def test_logger_optional_sinks(tmp_path, monkeypatch):
    """use_tensorboard / use_wandb (reference logger.py:130-150, 312-318): the row reaches the sink when its package
    is importable; when it is not, the logger WARNS and carries on with progress.csv (never a silent no-op)."""
    import sys
    import types
    import warnings

    from omnisafe_amd.logger import Logger

    calls = []

    class _Writer:
        def __init__(self, log_dir):
            calls.append(('init', log_dir))

        def add_scalar(self, key, val, global_step):
            calls.append(('scalar', key, val, global_step))

        def flush(self):
            pass

        def close(self):
            calls.append(('close',))

    mod = types.ModuleType('torch.utils.tensorboard.writer')
    mod.SummaryWriter = _Writer
    pkg = types.ModuleType('torch.utils.tensorboard')
    pkg.writer = mod
    monkeypatch.setitem(sys.modules, 'torch.utils.tensorboard', pkg)
    monkeypatch.setitem(sys.modules, 'torch.utils.tensorboard.writer', mod)
    lg = Logger(str(tmp_path / 'a'), 'exp', seed=0, use_tensorboard=True, verbose=False)
    lg.register_key('Loss/Loss_pi')
    lg.store({'Loss/Loss_pi': 2.0})
    lg.dump_tabular()
    lg.close()
    assert calls[0] == ('init', os.path.join(lg.log_dir, 'tb'))
    assert ('scalar', 'Loss/Loss_pi', 2.0, 0) in calls and calls[-1] == ('close',)
    # missing packages: a warning per requested sink, csv still written
    monkeypatch.setitem(sys.modules, 'torch.utils.tensorboard', None)
    monkeypatch.setitem(sys.modules, 'torch.utils.tensorboard.writer', None)
    monkeypatch.setitem(sys.modules, 'wandb', None)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        lg = Logger(str(tmp_path / 'b'), 'exp', seed=0, use_tensorboard=True, use_wandb=True, verbose=False)
    msgs = [str(w.message) for w in rec]
    assert any('use_tensorboard=True' in m for m in msgs) and any('use_wandb=True' in m for m in msgs)
    lg.register_key('Loss/Loss_pi')
    lg.store({'Loss/Loss_pi': 1.0})
    lg.dump_tabular()
    lg.close()
    assert len(list(csv.reader(open(os.path.join(lg.log_dir, 'progress.csv'))))) == 2


def test_logger_columns(tmp_path):
    from omnisafe_amd.logger import Logger

    lg = Logger(str(tmp_path), 'exp', seed=3, verbose=False)
    lg.register_key('Metrics/EpRet', window_length=3)
    lg.register_key('Train/PolicyRatio', min_and_max=True)
    lg.register_key('Loss/Loss_pi', delta=True)
    for v in (1.0, 2.0, 3.0, 4.0):
        lg.store({'Metrics/EpRet': v})
    lg.store({'Train/PolicyRatio': 0.5})
    lg.store({'Train/PolicyRatio': 1.5})
    lg.store({'Loss/Loss_pi': 2.0})
    assert lg.get_stats('Metrics/EpRet')[0] == pytest.approx(3.0)  # window of 3
    lg.dump_tabular()
    lg.store({'Loss/Loss_pi': 0.5})
    lg.store({'Train/PolicyRatio': 1.0})
    lg.dump_tabular()
    lg.close()
    rows = list(csv.reader(open(os.path.join(lg.log_dir, 'progress.csv'))))
    assert rows[0] == ['Metrics/EpRet', 'Train/PolicyRatio', 'Train/PolicyRatio/Min',
                       'Train/PolicyRatio/Max', 'Train/PolicyRatio/Std', 'Loss/Loss_pi',
                       'Loss/Loss_pi/Delta']
    # /Min and /Max as the reference WRITES them: the mean (element-wise reduction across ranks, then .mean():
    # distributed.py:388-390, logger.py:366; checked live against the reference's Logger in
    # test_oracle_vs_reference.py::test_logger_rows_live)
    assert float(rows[1][1]) == 1.0 and float(rows[1][2]) == 1.0 and float(rows[1][3]) == 1.0
    assert float(rows[1][4]) == 0.5
    assert float(rows[2][6]) == pytest.approx(-1.5)  # delta vs the previous epoch
    assert float(rows[2][0]) == pytest.approx(3.0)   # window keys persist across epochs


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_agent_fails_loudly_without_gpu():
    import omnisafe_amd

    with pytest.raises(RuntimeError, match='no CPU fallback|GPU only'):
        omnisafe_amd.Agent('PPOLag', 'SynthPointGoal1-v0')
    with pytest.raises(RuntimeError, match='GPU only'):
        omnisafe_amd.Agent('PPOLag', 'SynthPointGoal1-v0', custom_cfgs={'train_cfgs': {'device': 'cpu'}})


@pytest.mark.reference
@pytest.mark.skipif(torch.cuda.is_available(), reason='build-container wiring test')
def test_plugin_installs_behind_reference_agent(tmp_path, monkeypatch):
    """``import omnisafe; omnisafe_amd.install()``: the reference's own Agent/AlgoWrapper + YAML config
    machinery constructs OUR PPOLag class (and would run the HIP path on a GPU box)."""
    import ref_harness

    omnisafe = ref_harness.import_reference()
    import omnisafe_amd
    from omnisafe.algorithms import registry as ref_registry

    keep = dict(ref_registry.REGISTRY._module_dict)
    # the reference itself calls torch.cuda.set_device for a cuda device (algo_wrapper.py:164); there is
    # no GPU in the build container, so neutralise that one call to reach the registry lookup.
    monkeypatch.setattr(torch.cuda, 'set_device', lambda *_a, **_k: None)
    try:
        swapped = omnisafe_amd.install()
        assert 'PPOLag' in swapped
        reg_cls = ref_registry.REGISTRY.get('PPOLag')
        # "class named PPOLag, subclass of the reference PPOLag so isinstance and hook order hold" --
        # and our implementation first in the MRO, so every hook of the path resolves to the HIP side
        assert reg_cls.__name__ == 'PPOLag' and issubclass(reg_cls, keep['PPOLag'])
        assert reg_cls.__mro__[1] is omnisafe_amd.algorithms.registry.get('PPOLag')
        for hook in ('__init__', '_init_env', '_init_model', '_init', '_init_log', 'learn', '_update'):
            owner = next(c for c in reg_cls.__mro__ if hook in vars(c))
            assert owner.__module__.startswith('omnisafe_amd.'), (hook, owner)
        cfg = {'train_cfgs': {'device': 'cuda:0', 'total_steps': 2000, 'vector_env_nums': 4},
               'algo_cfgs': {'steps_per_epoch': 1000},
               'logger_cfgs': {'use_wandb': False, 'use_tensorboard': False, 'log_dir': str(tmp_path)}}
        assert len(swapped) == 23  # every on-policy algorithm of the reference
        for name in swapped:  # every accelerated algorithm is reachable through the reference's own Agent
            cls = ref_registry.REGISTRY.get(name)
            assert cls.__mro__[1] is omnisafe_amd.algorithms.registry.get(name) and issubclass(cls, keep[name])
            assert omnisafe_amd.install() == swapped and ref_registry.REGISTRY.get(name) is cls  # idempotent
            if name.endswith('EarlyTerminated'):  # utils/config.py:292-295: single env only
                c1 = dict(cfg, train_cfgs=dict(cfg['train_cfgs'], vector_env_nums=1))
            else:
                c1 = cfg
            with pytest.raises(RuntimeError, match='no CPU fallback'):
                omnisafe.Agent(name, 'SynthPointGoal1-v0', custom_cfgs=c1)
    finally:
        ref_registry.REGISTRY._module_dict.clear()
        ref_registry.REGISTRY._module_dict.update(keep)


@pytest.mark.parametrize('c,q,r,s', [(-1.0, 1.0, 1.0, 1.0), (-0.01, 1.0, 1.0, 1.0), (0.01, 1.0, 1.0, 1.0),
                                     (1.0, 1.0, 1.0, 1.0), (-0.05, 0.5, -0.2, 0.3), (0.05, 0.4, 0.1, 0.2),
                                     (0.3, 0.4, -0.1, 0.2), (-2.0, 0.4475, -0.0214, 0.2417)])
def test_cpo_case_algebra_matches_oracle(c, q, r, s):
    """CPO._determine_case / _step_direction host algebra (cpo.py:237-337): every optimisation case."""
    from omnisafe_amd.algorithms.trust_region_algos import cpo_determine_case, cpo_step_coefficients

    t = torch.tensor
    P = 12
    torch.manual_seed(0)
    x, p, b = torch.randn(P), torch.randn(P), torch.randn(P)
    for bvec in (b, b * 1e-4):
        case_o, A_o, B_o = O.cpo_determine_case(bvec, t(c), t(q), t(r), t(s), target_kl=0.01)
        case, A, B = cpo_determine_case(float(bvec.dot(bvec)), c, q, r, s, 0.01)
        assert case == case_o
        d_o, lam_o, nu_o = O.cpo_step_direction(case_o, t(q), x, A_o, B_o, t(q), p, t(r), t(s), t(c), 0.01)
        cx, cp, lam, nu = cpo_step_coefficients(case, q, A, B, q, r, s, c, 0.01)
        np.testing.assert_allclose((float(cx) * x + float(cp) * p).numpy(), d_o.numpy(), rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(float(lam), float(lam_o), rtol=2e-5)
        np.testing.assert_allclose(float(nu), float(nu_o), rtol=2e-5, atol=1e-7)


def test_default_configs_match_reference_yaml(golden_dir):
    """omnisafe_amd.config.DEFAULTS vs the snapshot of the reference's YAML `defaults` blocks
    (tests/golden/config_defaults.json, written by oracle/make_golden.py).  Deliberate deviations:
    device (cuda:0), use_tensorboard (off: tensorboard is optional here) and the `verbose` extension."""
    import json
    import os

    from omnisafe_amd import config

    ref_all = json.load(open(os.path.join(golden_dir, 'config_defaults.json')))

    def cmp(a, r, m, path=''):
        for k in r:
            if k == 'env_cfgs':
                continue
            assert k in m, f'{a}: missing {path}{k}'
            if isinstance(r[k], dict):
                cmp(a, r[k], m[k], path + k + '.')
            elif k not in ('device', 'use_tensorboard'):
                assert r[k] == m[k], f'{a}: {path}{k}: reference {r[k]!r}, ours {m[k]!r}'
        for k in m:
            assert k in r or k in ('env_cfgs', 'verbose'), f'{a}: extra {path}{k}'

    assert len(config.DEFAULTS) >= 14
    for algo in config.DEFAULTS:
        cmp(algo, ref_all[algo], config.get_default_kwargs(algo))


class _HostVecEnv:
    """Deterministic host vector env with the reference's CMDP surface (envs/core.py:37-182): CPU tensors out,
    gymnasium's vector auto-reset convention (final_observation + boolean numpy mask) every `horizon` steps for
    the even envs and every 2 * horizon for the odd ones."""
    need_auto_reset_wrapper = False
    need_time_limit_wrapper = False
    need_evaluation = False
    env_spec_log = {}

    def __init__(self, n=6, d_o=5, d_a=2, horizon=3):
        from omnisafe_amd.spaces import Box

        self.num_envs, self.d_o, self.d_a, self.h = n, d_o, d_a, horizon
        self.observation_space = Box(-np.inf, np.inf, (d_o,))
        self.action_space = Box(-2.0, 2.0, (d_a,))
        self.k = np.zeros(n, np.int64)
        self.t = 0
        self.actions = []

    def set_seed(self, seed):
        self.seed = seed

    def _obs(self):
        return torch.from_numpy((self.k[:, None] + 0.1 * np.arange(self.d_o)[None] + 100.0 * self.t).astype(np.float32))

    def reset(self, seed=None, options=None):
        self.k[:] = 0
        return self._obs(), {'why': 'reset'}

    def step(self, action):
        assert action.device.type == 'cpu' and tuple(action.shape) == (self.num_envs, self.d_a)
        self.actions.append(action.clone())
        self.t += 1
        self.k += 1
        lim = np.where(np.arange(self.num_envs) % 2 == 0, self.h, 2 * self.h)
        trunc = self.k >= lim
        obs = self._obs()
        info = {'goal_met': False}
        if trunc.any():
            info['final_observation'] = obs.clone()
            info['_final_observation'] = trunc.copy()  # numpy bool array, as gymnasium's vector envs
            self.k[trunc] = 0
            obs = self._obs()
        reward = torch.from_numpy(action.numpy().sum(1).astype(np.float32))
        cost = torch.from_numpy((self.k % 2).astype(np.float32))
        return obs, reward, cost, torch.zeros(self.num_envs, dtype=torch.bool), torch.from_numpy(trunc), info

    def close(self):
        self.closed = True


def test_host_env_bridge_staging_cpu():
    """HostEnvBridge (omnisafe_amd/host_env.py) on CPU tensors: what comes out of step() is what the host env
    returned, final rows travel only on steps where an env finished, byte counts are the documented
    4 D_a down and 4 (D_o + 5) (+ 4 D_o) up per env-step."""
    from omnisafe_amd.host_env import HostEnvBridge

    host = _HostVecEnv()
    twin = _HostVecEnv()
    br = HostEnvBridge(host, 'cpu')
    assert br.num_envs == 6 and br.observation_space.shape == (5,) and br.host_resident
    assert not br.need_auto_reset_wrapper and not br.need_time_limit_wrapper and not br.graph_safe
    o, info = br.reset()
    o2, _ = twin.reset()
    assert torch.equal(o, o2) and info == {'why': 'reset'}
    finals = 0
    for t in range(7):
        act = torch.full((6, 2), 0.25 * t)
        obs, r, c, term, trunc, info = br.step(act)
        eo, er, ec, eterm, etrunc, einfo = twin.step(act)
        assert torch.equal(obs, eo) and torch.equal(r, er) and torch.equal(c, ec)
        assert torch.equal(term, eterm.float()) and torch.equal(trunc, etrunc.float())
        assert ('final_observation' in info) == ('final_observation' in einfo)
        assert info['goal_met'] is False
        if 'final_observation' in einfo:
            finals += 1
            m = torch.from_numpy(einfo['_final_observation'])
            assert torch.equal(info['_final_observation'], m.float())
            assert torch.equal(info['final_observation'][m], einfo['final_observation'][m])
    assert finals == 2  # step 3 (even envs) and step 6 (all envs)
    down, up = br.pcie_bytes_per_env_step()
    assert down == 4 * 2
    n = 6
    want_up = (7 * 4 * (5 + 5) * n + finals * 4 * 5 * n) / (7 * n)
    assert up == pytest.approx(want_up + 4 * 5 * n / (7 * n))  # + the reset's observation upload
    br.set_seed(5)
    br.close()
    assert host.seed == 5 and host.closed


class _HostSingleEnv:
    """Single host env that asks for both wrappers (the shape of SafetyGymnasiumEnv with num_envs = 1,
    safety_gymnasium_env.py:147-158): scalar reward / cost / flags, terminates itself at step 4 of odd episodes."""
    need_auto_reset_wrapper = True
    need_time_limit_wrapper = True
    need_evaluation = False
    num_envs = 1
    max_episode_steps = 6

    def __init__(self):
        from omnisafe_amd.spaces import Box

        self.observation_space = Box(-np.inf, np.inf, (3,))
        self.action_space = Box(-1.0, 1.0, (2,))
        self.k, self.ep = 0, -1

    def set_seed(self, seed):
        pass

    def reset(self, seed=None, options=None):
        self.k, self.ep = 0, self.ep + 1
        return torch.tensor([float(self.k), float(self.ep), 0.5]), {}

    def step(self, action):
        assert tuple(action.shape) == (2,)  # squeezed, as behind the reference's Unsqueeze wrapper
        self.k += 1
        term = self.ep % 2 == 1 and self.k >= 4
        return (torch.tensor([float(self.k), float(self.ep), 0.5]), torch.tensor(1.0), torch.tensor(0.25),
                torch.tensor(term), torch.tensor(False), {})

    def close(self):
        pass


def test_host_env_bridge_single_env_wrappers_cpu():
    """A single host env behind the bridge: TimeLimit + AutoReset run on the host side (online_adapter.py:120-132),
    outputs are unsqueezed to N = 1 rows, the episode's true last observation arrives as final_observation."""
    from omnisafe_amd.host_env import AutoReset, HostEnvBridge, TimeLimit

    br = HostEnvBridge(_HostSingleEnv(), 'cpu')
    assert isinstance(br.host_env, AutoReset) and isinstance(br.host_env._env, TimeLimit)
    obs, _ = br.reset()
    assert obs.tolist() == [[0.0, 0.0, 0.5]]
    log = []
    for _ in range(11):
        obs, r, c, term, trunc, info = br.step(torch.zeros(1, 2))
        assert tuple(obs.shape) == (1, 3) and tuple(r.shape) == (1,) and float(r[0]) == 1.0 and float(c[0]) == 0.25
        log.append((obs[0, 0].item(), obs[0, 1].item(), bool(term[0]), bool(trunc[0]),
                    info['final_observation'][0, 0].item() if 'final_observation' in info else None))
    # episode 0: 6 steps, truncated by the time limit; episode 1: terminates itself at its 4th step
    assert log[5] == (0.0, 1.0, False, True, 6.0)
    assert log[9] == (0.0, 2.0, True, False, 4.0)
    assert [x[4] for x in log[:5]] == [None] * 5 and log[10][:2] == (1.0, 2.0)


def test_model_cfgs_activation_and_width_checks():
    """models.py: activation names of the reference (utils/model.py:47-70) map to the ABI's codes; unknown activations,
    an unknown actor type and more hidden layers than the descriptor holds raise NotImplementedError BEFORE the
    library is touched (as the reference raises for unknown names).  Round 4: every hidden_sizes list the reference's
    builder accepts is taken -- the shapes outside the fused [H, H] family go to the layer-wise path
    (csrc/general_mlp.hip) and fail here, in the GPU-less container, only at the library's GPU requirement."""
    import types

    from omnisafe_amd import _lib, models

    assert models.ACTIVATIONS == {'tanh': 0, 'relu': 1, 'sigmoid': 2, 'softplus': 3, 'identity': 4}
    src = open(models.__file__).read()
    assert "self.hidden = self.width | (ACTIVATIONS[self.activation] << 16)" in src
    ns = types.SimpleNamespace
    from omnisafe_amd.spaces import Box

    def build(a_act, c_act, a_hid, c_hid, actor_type='gaussian_learning'):
        cfg = ns(actor=ns(hidden_sizes=a_hid, activation=a_act, lr=3e-4),
                 critic=ns(hidden_sizes=c_hid, activation=c_act, lr=3e-4),
                 weight_initialization_mode='kaiming_uniform', actor_type=actor_type, linear_lr_decay=True)
        return models.ConstraintActorCritic(Box(-1, 1, (4,)), Box(-1, 1, (2,)), cfg, 2, device='cuda:0')

    for args in (('gelu', 'gelu', [64, 64], [64, 64]), ('tanh', 'swish', [64, 64], [64, 64]),
                 ('tanh', 'tanh', [64] * 8, [64] * 8), ('tanh', 'tanh', [64, 0], [64, 64]),
                 ('tanh', 'tanh', [64, 64], [64, 64], 'mlp')):
        with pytest.raises(NotImplementedError):
            build(*args)
    if not torch.cuda.is_available():  # accepted shapes get as far as the library's GPU requirement
        for args in (('relu', 'tanh', [64, 64], [64, 64]), ('tanh', 'tanh', [96, 96], [96, 96]),
                     ('tanh', 'tanh', [64, 32], [64, 32]), ('tanh', 'tanh', [64, 64, 64], [64]),
                     ('tanh', 'tanh', [1024, 1024], [1024, 1024]), ('tanh', 'tanh', [64, 64], [64, 64])):
            with pytest.raises((_lib.OsaError, RuntimeError)) as ei:
                build(*args)
            assert not isinstance(ei.value, NotImplementedError)


def test_shuffle_twin_is_a_bijection_for_every_size():
    """oracle/np_oracle.py:shuffle_rows (the numpy twin of csrc/shuffle_kernels.hip): every row a permutation, also
    where the domain is not a power of two (cycle walking) and for one-element inputs; seeds matter."""
    import numpy as np

    import np_oracle as O

    for M in (1, 2, 3, 7, 8, 9, 100, 1000, 4097):
        p = O.shuffle_rows([5, 6, (1 << 62) - 1], M)
        for r in p:
            assert np.array_equal(np.sort(r), np.arange(M))
        if M > 8:
            assert not np.array_equal(p[0], p[1])
    assert np.array_equal(O.shuffle_rows([9], 50), O.shuffle_rows([9], 50))


@pytest.mark.reference
def test_plugin_passes_cpu_device_through_to_the_reference(tmp_path):
    """BASELINE config 1 ("PPOLag ..., torch CPU device (reference plumbing, no GPU)"): the reference's YAML default
    is `device: cpu` (configs/on-policy/PPOLag.yaml:22; utils/tools.py:338-358), so a default-config script must
    keep working after `omnisafe_amd.install()`: the registry entry dispatches on `cfgs.train_cfgs.device` and hands
    a CPU run to the SAVED reference class (algo_wrapper.py:150-170).  Same seed => the csv of the installed run
    equals the csv of an uninstalled run, column for column (wall-clock columns aside)."""
    import csv
    import glob

    import ref_harness

    omnisafe = ref_harness.import_reference()
    ref_harness.import_simple_env()  # the reference's own tests/simple_env.py ('Test-v0')
    import omnisafe_amd
    from omnisafe.algorithms import registry as ref_registry

    keep = dict(ref_registry.REGISTRY._module_dict)

    def run(tag):
        # no `device` key: the YAML default (cpu) decides
        cfg = {'train_cfgs': {'total_steps': 400, 'vector_env_nums': 1, 'torch_threads': 1},
               'algo_cfgs': {'steps_per_epoch': 200, 'update_iters': 2},
               'logger_cfgs': {'use_wandb': False, 'use_tensorboard': False, 'log_dir': str(tmp_path / tag)}}
        agent = omnisafe.Agent('PPOLag', 'Test-v0', custom_cfgs=cfg)
        assert agent.cfgs.train_cfgs.device == 'cpu'
        agent.learn()
        path, = glob.glob(str(tmp_path / tag / '**' / 'progress.csv'), recursive=True)
        rows = list(csv.DictReader(open(path)))
        return agent, rows

    try:
        plain_agent, plain = run('plain')
        assert 'PPOLag' in omnisafe_amd.install()
        agent, rows = run('installed')
        assert type(agent.agent) is keep['PPOLag'] is type(plain_agent.agent)  # the reference's class, untouched
        assert not type(agent.agent).__module__.startswith('omnisafe_amd')
        assert len(rows) == len(plain) == 2 and list(rows[0]) == list(plain[0])
        for a, b in zip(rows, plain):
            for k in a:
                if not k.startswith('Time/'):
                    assert a[k] == b[k], (k, a[k], b[k])
        # ... and a cuda request still reaches the HIP class (which refuses to run without a GPU: no CPU fallback)
        if not torch.cuda.is_available():
            import unittest.mock as mock

            with mock.patch.object(torch.cuda, 'set_device', lambda *_a, **_k: None):
                with pytest.raises(RuntimeError, match='no CPU fallback'):
                    omnisafe.Agent('PPOLag', 'Test-v0', custom_cfgs={'train_cfgs': {'device': 'cuda:0'}})
    finally:
        omnisafe_amd.uninstall()
        ref_registry.REGISTRY._module_dict.clear()
        ref_registry.REGISTRY._module_dict.update(keep)


@pytest.mark.reference
def test_plugin_cpu_parallel_goes_through_the_reference_fork(tmp_path):
    """`train_cfgs.parallel = 2` on the default CPU device with the plugin installed: the reference's own `fork`
    (utils/distributed.py:83-139) re-launches the script under torchrun; both workers get the reference's class from
    the swapped registry entry and the run's csv equals that of the same script without the plugin."""
    import csv
    import glob
    import shutil
    import subprocess

    if shutil.which('torchrun') is None:
        pytest.skip('torchrun not on PATH (the reference forks through it)')
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_plugin_parallel_cpu_script.py')
    rows = {}
    for tag, flag in (('plain', '0'), ('installed', '1')):
        env = dict(os.environ, WITH_PLUGIN=flag)
        for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'IN_DIST'):
            env.pop(k, None)
        p = subprocess.run([sys.executable, script, str(tmp_path / tag)], capture_output=True, text=True, env=env,
                           timeout=600, cwd=str(tmp_path))
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        classes = [ln for ln in p.stdout.splitlines() if ln.startswith('CLASS')]
        assert len(classes) == 2 and all('omnisafe.algorithms' in c for c in classes), classes  # two workers
        path, = glob.glob(str(tmp_path / tag / '**' / 'progress.csv'), recursive=True)  # rank 0 writes
        rows[tag] = list(csv.DictReader(open(path)))
    assert len(rows['plain']) == 2
    for a, b in zip(rows['installed'], rows['plain']):
        assert list(a) == list(b)
        for k in a:
            if not k.startswith('Time/'):
                assert a[k] == b[k], (k, a[k], b[k])


def test_logger_deferred_rows_equal_in_place_rows(tmp_path, monkeypatch):
    """Logger.dump_tabular() snapshots the epoch and the csv row is written by flush() (behind the next rollout's launch,
    close() at the latest): the same rows in the same order as with OSA_LOG_DEFER=0, at most one epoch later on disk."""
    import csv as _csv

    from omnisafe_amd.logger import Logger

    files = {}
    for defer in ('1', '0'):
        monkeypatch.setenv('OSA_LOG_DEFER', defer)
        lg = Logger(str(tmp_path / defer), 'x', verbose=False)
        lg.register_key('Loss')
        lg.register_key('Metrics/EpRet', window_length=3, min_and_max=True, delta=True)
        path = os.path.join(lg.log_dir, 'progress.csv')
        for e in range(4):
            lg.store({'Loss': 0.5 * e}, **{'Metrics/EpRet': float(e * e)})
            lg.store({'Loss': 0.25 * e})
            lg.dump_tabular()
            on_disk = len(open(path).read().splitlines())
            assert on_disk == (e + 1 if defer == '1' and e > 0 else (0 if defer == '1' else e + 2))
            assert lg.current_epoch == e + 1
            if e == 1:
                lg.flush()  # (what the adapter does once the next rollout is enqueued)
                assert len(open(path).read().splitlines()) == e + 2
        lg.close()
        files[defer] = list(_csv.reader(open(path)))
    assert files['1'] == files['0'] and len(files['1']) == 5



def test_bench_refuses_more_ranks_than_devices():
    """`python bench.py --gpus N` without torchrun's environment launches its own ranks (as the reference's `fork`,
    omnisafe/utils/distributed.py:121-137); on a box with fewer than N devices it refuses loudly (exit code 2, the test
    hook named) instead of stacking ranks on one device silently.  Runs here without a GPU: device_count() == 0."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'OSA_SINGLE_DEVICE_RANKS')}
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip('a multi-GPU box launches the ranks for real')
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2, (p.returncode, p.stderr[-500:])
    assert 'OSA_SINGLE_DEVICE_RANKS' in p.stderr and '--gpus 2' in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith('{')]  # no bench line from a refused launch
