"""Live cross-checks of the oracle (and of the product's pure-host classes) against the UNMODIFIED reference,
imported from /root/reference through oracle/ref_harness.py.  Build container only: skipped where the
reference is absent (e.g. the GPU box) -- there the committed golden vectors carry the same information."""
import os

import numpy as np
import pytest
import torch

import np_oracle as O

pytestmark = pytest.mark.reference


@pytest.fixture(scope='module')
def ref():
    import ref_harness

    return ref_harness.import_reference()


def test_discount_cumsum_live(ref):
    from omnisafe.utils.math import discount_cumsum

    rng = np.random.default_rng(0)
    for n, d in ((1, 0.99), (17, 0.95), (1000, 0.9405)):
        x = rng.standard_normal(n).astype(np.float32)
        want = discount_cumsum(torch.from_numpy(x), d).numpy()
        assert np.array_equal(O.discount_cumsum(x, d), want)


def test_onpolicy_buffer_paths_live(ref):
    """OnPolicyBuffer.store / finish_path / get of the reference vs the oracle's per-path GAE (bit-exact)."""
    from omnisafe.common.buffer import OnPolicyBuffer

    from ref_harness import _Box

    rng = np.random.default_rng(1)
    T, gamma, lam, lam_c = 37, 0.99, 0.95, 0.9
    buf = OnPolicyBuffer(_Box(-np.inf, np.inf, (3,)), _Box(-1, 1, (2,)), T, gamma, lam, lam_c, 'gae', 0.0,
                         False, False, device=torch.device('cpu'))
    d = {k: rng.standard_normal(T).astype(np.float32) for k in ('reward', 'cost', 'value_r', 'value_c', 'logp')}
    path_end = np.zeros(T, np.uint8)
    path_end[[9, 10, 25, T - 1]] = 1
    boot_r = np.where(path_end, rng.standard_normal(T), 0).astype(np.float32)
    boot_c = np.where(path_end, rng.standard_normal(T), 0).astype(np.float32)
    for t in range(T):
        buf.store(obs=torch.zeros(3), act=torch.zeros(2), **{k: torch.tensor(v[t]) for k, v in d.items()})
        if path_end[t]:
            buf.finish_path(torch.tensor([boot_r[t]]), torch.tensor([boot_c[t]]))
    want = {k: buf.data[k].numpy().copy() for k in ('adv_r', 'adv_c', 'target_value_r', 'target_value_c')}
    got = O.gae_per_path(d['reward'][:, None], d['cost'][:, None], d['value_r'][:, None], d['value_c'][:, None],
                         path_end[:, None], boot_r[:, None], boot_c[:, None], gamma, lam, lam_c)
    for ok, rk in (('adv_r', 'adv_r'), ('adv_c', 'adv_c'), ('tgt_r', 'target_value_r'), ('tgt_c', 'target_value_c')):
        assert np.array_equal(got[ok][:, 0], want[rk]), ok


def test_normalizer_live(ref):
    from omnisafe.common.normalizer import Normalizer

    rng = np.random.default_rng(2)
    a, b = Normalizer((5,), clip=5), O.Normalizer((5,), clip=5)
    for n in (1, 4, 64, 3):
        x = torch.from_numpy(rng.standard_normal((n, 5)).astype(np.float32) * 3 + 1)
        ya, yb = a.normalize(x.clone()), b.normalize(x.clone())
        assert torch.equal(ya, yb)
    for k in ('mean', 'sumsq', 'var', 'std'):
        assert torch.equal(getattr(a, '_' + k), getattr(b, k)), k
    assert int(a._count) == int(b.count)


def test_lagrange_and_pid_live(ref):
    """Naive Lagrange (oracle and the product's host class) and the product's PID controller vs the reference."""
    from omnisafe.common.lagrange import Lagrange as RefLag
    from omnisafe.common.pid_lagrange import PIDLagrangian as RefPID

    from omnisafe_amd.lagrange import Lagrange
    from omnisafe_amd.pid_lagrange import PIDLagrangian

    kw = dict(cost_limit=25.0, lagrangian_multiplier_init=0.001, lambda_lr=0.035, lambda_optimizer='Adam')
    r, o, p = RefLag(**kw), O.Lagrange(**kw), Lagrange(**kw)
    for jc in (30.0, 80.5, 10.0, 25.0, 60.25):
        for x in (r, o, p):
            x.update_lagrange_multiplier(jc)
        want = float(r.lagrangian_multiplier)
        assert float(o.lagrangian_multiplier) == want and p.lagrangian_multiplier == want
    pk = dict(pid_kp=0.1, pid_ki=0.01, pid_kd=0.01, pid_d_delay=10, pid_delta_p_ema_alpha=0.95,
              pid_delta_d_ema_alpha=0.95, sum_norm=True, diff_norm=False, penalty_max=100,
              lagrangian_multiplier_init=0.001, cost_limit=25.0)
    costs = (30.0, 80.5, 10.0, 25.0, 60.25, 26.0, 24.0, 90.0, 0.0, 33.0, 41.0, 12.0, 25.5, 700.0, 3.0, 28.0)
    # all three output ranges (sum_norm / diff_norm / penalty_max ceiling) and delay lines that wrap around
    for over in ({}, {'pid_d_delay': 3}, {'sum_norm': False, 'diff_norm': True, 'pid_d_delay': 2},
                 {'sum_norm': False, 'diff_norm': False, 'penalty_max': 2, 'pid_kp': 0.5, 'pid_d_delay': 1},
                 {'pid_ki': 0.2, 'pid_kd': 0.3, 'pid_delta_p_ema_alpha': 0.5, 'pid_delta_d_ema_alpha': 0.7,
                  'lagrangian_multiplier_init': 0.4}):
        kw_pid = dict(pk, **over)
        rp, pp = RefPID(**kw_pid), PIDLagrangian(**kw_pid)
        for jc in costs:
            rp.pid_update(jc)
            pp.pid_update(jc)
            assert rp.lagrangian_multiplier == pp.lagrangian_multiplier, (over, jc)


def test_conjugate_gradients_live(ref):
    from omnisafe.utils.math import conjugate_gradients

    torch.manual_seed(0)
    A = torch.randn(40, 40)
    A = A @ A.T + 0.5 * torch.eye(40)
    b = torch.randn(40)
    fvp = lambda v: A @ v  # noqa: E731
    assert torch.equal(conjugate_gradients(fvp, b, 15), O.conjugate_gradients(fvp, b, 15))


def test_simmer_controller_live(ref):
    """The product's SimmerPIDController vs the reference's SimmerPIDAgent (host tensors)."""
    import types

    from omnisafe.common.simmer_agent import SimmerPIDAgent

    from omnisafe_amd.adapter import SimmerPIDController

    cfgs = types.SimpleNamespace(kp=0.05, ki=0.01, kd=0.02, polyak=0.9)
    bound = 2.0 * torch.ones(4, 1)
    a, b = SimmerPIDAgent(cfgs=cfgs, budget_bound=bound), SimmerPIDController(cfgs, budget_bound=bound)
    sa = sb = torch.ones(4, 1)
    for jc in (1.5, 0.2, 3.0, 0.9, 2.5, 0.0):
        sa = a.act(safety_budget=sa, observation=torch.tensor(jc))
        sb = b.act(safety_budget=sb, observation=torch.tensor(jc))
        assert torch.equal(sa, sb)


def test_reach_env_twin_live(ref):
    """The CPU env the reference trains on for the learning curves (oracle/ref_harness.py:
    register_reach_env) behaves as a reference CMDP: spaces, vector auto-reset convention, horizon."""
    import ref_harness

    from omnisafe.envs.core import CMDP

    cls = ref_harness.register_reach_env()  # instantiated directly: omnisafe_amd.install() (another test)
    assert issubclass(cls, CMDP)            # may have pointed the id at the device env in this process
    env = cls('SynthReach-v0', num_envs=4, device=torch.device('cpu'), horizon=5)
    env.set_seed(3)
    obs, _ = env.reset()
    assert obs.shape == (4, 60) and env.action_space.shape == (2,)
    state0 = env._state.copy()
    for t in range(5):
        act = torch.full((4, 2), 0.5)
        q, r, c, reached = O.reach_env_step(env._state, act.numpy())
        obs, reward, cost, term, trunc, info = env.step(act)
        np.testing.assert_array_equal(reward.numpy(), r)
        np.testing.assert_array_equal(cost.numpy(), c)
        assert bool(trunc.all()) == (t == 4) and not bool(term.any())
    np.testing.assert_array_equal(info['final_observation'][:, 0:2].numpy(), q)
    assert not np.array_equal(env._state, state0) and np.all(np.abs(env._state) <= 1)


def test_learning_runs_get_the_same_custom_cfgs_on_both_sides(ref, tmp_path):
    """The learning-curve comparison feeds the reference (oracle/make_golden.py:learning_custom_cfgs, from the
    YAML defaults) and omnisafe_amd (tests/test_learning_gpu.py:reach_custom_cfgs, from omnisafe_amd's
    defaults) the same custom_cfgs for every algorithm, device aside."""
    import json

    import make_golden
    import test_learning_gpu as tl
    from omnisafe.utils.config import get_default_kwargs_yaml

    cfg = json.load(open(tl.GOLDEN))['config']
    assert cfg == make_golden.LEARNING_CFG
    for algo in make_golden.LEARNING_ALGOS:
        defaults = get_default_kwargs_yaml(algo, cfg['env_id'], 'on-policy').todict()
        theirs = make_golden.learning_custom_cfgs(algo, 3, 'cpu', str(tmp_path), defaults)
        ours = tl.reach_custom_cfgs(algo, 3, cfg, str(tmp_path))
        assert theirs['train_cfgs'].pop('device') == 'cpu' and ours['train_cfgs'].pop('device') == tl.DEV
        assert theirs == ours, algo
    assert sorted(make_golden.LEARNING_ALGOS) == sorted(['PPOLag', 'TRPOLag', 'CPO'] + tl.SIBLINGS)


def test_logger_rows_live(ref, tmp_path, monkeypatch):
    """The same sequence of register / store / dump calls through the reference's Logger and ours: identical
    progress.csv (header and values, incl. the /Min /Max columns, which the reference fills with the mean,
    window keys, /Delta and /Std)."""
    import csv

    monkeypatch.setenv('OMNISAFE_DEVICE', 'cpu')  # (the reference's AlgoWrapper exports the device of the last Agent)
    from omnisafe.common.logger import Logger as RefLogger
    from omnisafe.utils.config import Config

    from omnisafe_amd.logger import Logger

    def drive(lg):
        lg.register_key('Metrics/EpRet', window_length=3)
        lg.register_key('Train/PolicyRatio', min_and_max=True)
        lg.register_key('Loss/Loss_pi', delta=True)
        lg.register_key('Metrics/LagrangeMultiplier', min_and_max=True)
        rng = np.random.default_rng(0)
        for epoch in range(3):
            for v in rng.normal(size=5):
                lg.store({'Metrics/EpRet': float(v)})
            for v in rng.normal(1.0, 0.1, size=7):
                lg.store({'Train/PolicyRatio': float(v)})
            lg.store({'Loss/Loss_pi': float(rng.normal())})
            lg.store({'Metrics/LagrangeMultiplier': 0.1 * epoch})
            lg.dump_tabular()
        lg.close()
        return list(csv.reader(open(os.path.join(lg.log_dir, 'progress.csv'))))

    cfg = Config.dict2config({'exp_name': 'x', 'seed': 0, 'logger_cfgs': {'use_wandb': False, 'use_tensorboard': False}})
    ref_rows = drive(RefLogger(str(tmp_path / 'ref'), 'exp', seed=0, use_tensorboard=False, use_wandb=False,
                               config=cfg))
    our_rows = drive(Logger(str(tmp_path / 'ours'), 'exp', seed=0, verbose=False))
    assert our_rows[0] == ref_rows[0]
    np.testing.assert_allclose(np.array(our_rows[1:], dtype=float), np.array(ref_rows[1:], dtype=float), rtol=1e-6,
                               atol=1e-7)
    k = ref_rows[0].index('Train/PolicyRatio')
    assert ref_rows[1][k + 1] == ref_rows[1][k + 2]  # the reference's Min == Max (== mean)


def test_hidden_shape_golden_is_what_the_generator_produces(ref, tmp_path):
    """tests/golden/hidden96x40x24_p3o_point.npz -- one whole `_update()` of the UNMODIFIED reference built with
    hidden_sizes outside the [64, 64] family (oracle/make_golden.py::gen_hidden_shape_updates) -- regenerated from the
    reference checkout by the committed script (own process: no plugin of this package installed in it): every array of
    the committed fixture comes out again, bit for bit."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    golden = os.path.join(root, 'tests', 'golden', 'hidden96x40x24_p3o_point.npz')
    env = dict(os.environ, OSA_GOLDEN_OUT=str(tmp_path))
    p = subprocess.run([sys.executable, os.path.join(root, 'oracle', 'make_golden.py'), 'hidden-shapes',
                        'hidden96x40x24_p3o_point'], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    new, old = np.load(tmp_path / 'hidden96x40x24_p3o_point.npz'), np.load(golden)
    assert sorted(new.files) == sorted(old.files)
    differing = [k for k in old.files if not (new[k].shape == old[k].shape and (
        np.array_equal(new[k], old[k]) if new[k].dtype.kind in 'iufb' else str(new[k]) == str(old[k])))]
    assert not differing, differing
