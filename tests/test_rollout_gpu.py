"""GPU parity of the rollout side: running normaliser, the whole OnPolicyAdapter.rollout replayed on
the reference's recorded environment trace, the synthetic device env, and an end-to-end
``Agent('PPOLag').learn()``.

Tolerances: normaliser statistics rtol 1e-5 (float64 batch moments vs the reference's float32);
stored observations / actions / values / logp rtol 1e-4, atol 2e-5 (normalised obs feed float32 MFMA
chains); GAE outputs rtol 1e-4, atol 1e-4 (they inherit the value differences); episode metrics and
path boundaries exact."""
import csv
import glob
import os
import types

import numpy as np
import pytest
import torch

import np_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_normalizer_vs_reference(golden):
    from omnisafe_amd.normalizer import Normalizer

    g = golden('normalizer.npz')
    norm = Normalizer((7,), clip=5, device=DEV)
    for i in range(int(g['n_batches'])):
        y = norm.normalize(torch.from_numpy(g[f'in{i}']).to(DEV))
        np.testing.assert_allclose(y.cpu().numpy(), g[f'out{i}'], rtol=2e-5, atol=1e-6, err_msg=str(i))
        np.testing.assert_allclose(norm.mean.cpu().numpy(), g[f'mean{i}'], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(norm._sumsq.cpu().numpy(), g[f'sumsq{i}'], rtol=1e-5, atol=1e-7)
        if i > 0:
            np.testing.assert_allclose(norm.std.cpu().numpy(), g[f'std{i}'], rtol=1e-5)
        else:
            assert np.isnan(norm.std.cpu().numpy()).all()  # count == 1: 0/0, as the reference
        assert int(norm._count) == int(g[f'count{i}'])
    sd = norm.state_dict()
    assert list(sd) == ['_mean', '_sumsq', '_var', '_std', '_count', '_clip']  # Evaluator-loadable


def test_normalizer_masked_push_matches_oracle():
    from omnisafe_amd.normalizer import Normalizer

    rng = np.random.default_rng(5)
    ref = O.Normalizer((60,), clip=5)
    norm = Normalizer((60,), clip=5, device=DEV)
    for _ in range(4):
        x = (rng.standard_normal((300, 60)) * 3 + 1).astype(np.float32)
        mask = rng.random(300) < 0.3
        yr = ref.normalize(torch.from_numpy(x[mask]))
        y = norm.normalize(torch.from_numpy(x).to(DEV), mask=torch.from_numpy(mask).to(DEV))
        np.testing.assert_allclose(y.cpu().numpy()[mask], yr.numpy(), rtol=1e-4, atol=2e-5)
        assert np.array_equal(y.cpu().numpy()[~mask], x[~mask])  # unselected rows pass through
        x2 = rng.standard_normal((300, 60)).astype(np.float32)
        np.testing.assert_allclose(norm.normalize(torch.from_numpy(x2).to(DEV)).cpu().numpy(),
                                   ref.normalize(torch.from_numpy(x2)).numpy(), rtol=1e-4, atol=2e-5)
    assert int(norm._count) == ref.count
    # an all-false mask is a no-op (the reference only pushes when some env finished)
    before = norm.mean.clone()
    norm.push(torch.zeros(300, 60, device=DEV), mask=torch.zeros(300, dtype=torch.uint8, device=DEV))
    assert torch.equal(before, norm.mean)


def test_normalizer_batch_sizes_share_one_workspace():
    """One Normalizer, batches of very different sizes (the reference accepts a (4096, D) batch followed by a
    single (D,) observation): the arrival ticket of the single-launch reduction must not move with N -- round 2
    kept it BEHIND the partial sums, where a later small batch found a leftover partial and never merged."""
    from omnisafe_amd.normalizer import Normalizer

    rng = np.random.default_rng(11)
    ref = O.Normalizer((60,), clip=5)
    norm = Normalizer((60,), clip=5, device=DEV)
    for n in (4096, 1, 300, 4096, 7, 1):
        x = (rng.standard_normal((n, 60)) * 2 - 0.5).astype(np.float32)
        xin = x[0] if n == 1 else x  # a single (D,) observation, as Evaluator / single-env adapters pass it
        y = norm.normalize(torch.from_numpy(xin).to(DEV))
        yr = ref.normalize(torch.from_numpy(xin))
        assert tuple(y.shape) == tuple(yr.shape)
        np.testing.assert_allclose(y.cpu().numpy(), yr.numpy(), rtol=1e-4, atol=2e-5)
        assert int(norm._count) == ref.count
    np.testing.assert_allclose(norm.mean.cpu().numpy(), ref.mean.numpy(), rtol=1e-5, atol=1e-6)


class TraceEnv:
    """Replays the raw env outputs recorded from the reference run (tests/golden/ppolag_epoch.npz)."""
    need_auto_reset_wrapper = False
    need_time_limit_wrapper = False

    def __init__(self, g, dev=None):
        from omnisafe_amd.spaces import Box

        self.dev = dev or DEV  # 'cpu': a HOST env (driven through omnisafe_amd.host_env.HostEnvBridge)
        self.g, self.t = g, 0
        self.num_envs = int(g['N'])
        self.observation_space = Box(-np.inf, np.inf, (60,))
        self.action_space = Box(-1.0, 1.0, (2,))
        self.actions = []

    def set_seed(self, seed):
        pass

    def reset(self, seed=None, options=None):
        self.t = 0
        return torch.from_numpy(self.g['rollout/reset_obs']).to(self.dev), {}

    def step(self, action):
        g, t = self.g, self.t
        assert action.device.type == torch.device(self.dev).type
        self.actions.append(action.cpu().numpy().copy())
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)  # noqa: E731
        info = {}
        fin = g['rollout/truncated'][t] | g['rollout/terminated'][t]
        if fin.any():
            info['final_observation'] = dev(g['rollout/final_obs'][t])
            info['_final_observation'] = dev(fin.astype(np.uint8))
        self.t += 1
        return (dev(g['rollout/obs'][t]), dev(g['rollout/reward'][t]), dev(g['rollout/cost'][t]),
                dev(g['rollout/terminated'][t].astype(np.uint8)),
                dev(g['rollout/truncated'][t].astype(np.uint8)), info)

    def close(self):
        pass


class _LoggerStub:
    def __init__(self):
        self.data = {}
        self._headers_windows = {'Metrics/EpRet': 100}

    def window_length(self, key):
        return self._headers_windows.get(key)

    def extend(self, k, v):
        self.data.setdefault(k, []).extend(v)

    def store(self, d):
        for k, v in d.items():
            self.data.setdefault(k, []).append(v)


def _cfgs(**algo):
    ns = types.SimpleNamespace
    a = dict(obs_normalize=True, reward_normalize=False, cost_normalize=False, use_cost=True)
    a.update(algo)
    return ns(train_cfgs=ns(device=DEV), algo_cfgs=ns(**a), env_cfgs=None)


@pytest.mark.parametrize('host_env', [False, True], ids=['device-env', 'host-env-bridge'])
def test_rollout_on_reference_trace(golden, host_env):
    """The whole rollout on the env trace recorded from the unmodified reference, noise injected: buffer rows,
    normaliser state and episode metrics equal the reference's.  `host-env-bridge`: the SAME trace served by a host
    env (CPU tensors in and out, as Safety-Gymnasium behind envs/safety_gymnasium_env.py:190-210) through
    HostEnvBridge -- one D2H / H2D pair per vector step, everything else on the device, identical rows."""
    from omnisafe_amd.adapter import HostEnvBridge, OnPolicyAdapter
    from omnisafe_amd.buffer import VectorOnPolicyBuffer
    from test_mlp_gpu import make_ac

    g = golden('ppolag_epoch.npz')
    N, T = int(g['N']), int(g['T'])
    env = TraceEnv(g, dev='cpu' if host_env else DEV)
    adapter = OnPolicyAdapter('trace', N, 0, _cfgs(), env=HostEnvBridge(env, DEV) if host_env else env)
    ac = make_ac(60, 2, g, 'init/')
    eps_iter = iter(g['rollout/eps'])
    plain_step = ac.step

    def step_with_recorded_noise(obs, deterministic=False, eps=None, out=None, nets_mask=7):
        if out is not None and 'act' in out and not deterministic:  # the vector policy step
            eps = torch.from_numpy(next(eps_iter)).to(DEV)
        return plain_step(obs, deterministic=deterministic, eps=eps, out=out, nets_mask=nets_mask)

    ac.step = step_with_recorded_noise
    buf = VectorOnPolicyBuffer(adapter.observation_space, adapter.action_space, T, 0.99, 0.95, 0.95,
                               'gae', 0.0, True, True, num_envs=N, device=DEV)
    logger = _LoggerStub()
    adapter.rollout(T, ac, buf, logger)
    assert buf.ptr == T
    buf.compute_advantages()
    b = {k: v.cpu().numpy() for k, v in buf.data.items()}
    for k in ('obs', 'act', 'value_r', 'value_c', 'logp'):
        np.testing.assert_allclose(b[k], g[f'buffer/{k}'], rtol=1e-4, atol=2e-5, err_msg=k)
    assert np.array_equal(b['reward'], g['buffer/reward']) and np.array_equal(b['cost'], g['buffer/cost'])
    # the env saw the ActionScale'd actions the reference's env saw
    np.testing.assert_allclose(np.stack(env.actions), g['rollout/action'], rtol=1e-4, atol=2e-5)
    # path boundaries: truncation every `horizon` steps and at epoch end
    pe = (g['rollout/truncated'] | g['rollout/terminated']).astype(np.uint8)
    pe[-1] = 1
    assert np.array_equal(b['path_end'], pe)
    for k in ('adv_r', 'adv_c', 'target_value_r', 'target_value_c', 'discounted_ret'):
        np.testing.assert_allclose(b[k], g[f'buffer/{k}'], rtol=1e-4, atol=1e-4, err_msg=k)
    norm = adapter.save()['obs_normalizer']
    np.testing.assert_allclose(norm.mean.cpu().numpy(), g['rollout/norm_mean'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(norm.std.cpu().numpy(), g['rollout/norm_std'], rtol=1e-5)
    assert int(norm._count) == int(g['rollout/norm_count'])
    # episode metrics in the reference's (step, env) order
    np.testing.assert_allclose(logger.data['Metrics/EpRet'], g['rollout/ep_ret_window'], rtol=1e-6)
    assert np.array_equal(np.float32(logger.data['Metrics/EpCost']), g['rollout/ep_cost_window'])
    assert np.array_equal(np.float32(logger.data['Metrics/EpLen']), g['rollout/ep_len_window'])
    np.testing.assert_allclose(logger.data['Value/reward'][0], g['rollout/value_r_log_mean'], rtol=1e-3,
                               atol=1e-5)
    if host_env:
        # PCIe bytes per env-step: 4 D_a down; 4 (D_o + 5) up, + 4 D_o on steps where an env finished
        br = adapter._env
        down, up = br.pcie_bytes_per_env_step()
        n_final = int(((g['rollout/truncated'] | g['rollout/terminated']).any(axis=1)).sum())
        assert down == 4 * 2
        assert up == pytest.approx((T * 4 * 65 + n_final * 4 * 60 + 4 * 60) / T)  # (+ the reset's observation)
        assert not adapter.last_rollout_graphed if hasattr(adapter, 'last_rollout_graphed') else True


def test_synth_env_statistics_and_autoreset():
    from omnisafe_amd import envs

    env = envs.make('SynthPointGoal1-v0', num_envs=4096, device=DEV, horizon=5, cost_p=0.05)
    env.set_seed(3)
    obs0, _ = env.reset()
    assert obs0.shape == (4096, 60)
    costs, rewards, all_obs = [], [], [obs0.clone()]
    for t in range(1, 11):
        obs, r, c, term, trunc, info = env.step(torch.zeros(4096, 2, device=DEV))
        all_obs.append(obs.clone())
        costs.append(c.clone())
        rewards.append(r.clone())
        assert int(term.sum()) == 0
        if t % 5 == 0:
            assert int(trunc.sum()) == 4096 and 'final_observation' in info
            assert info['final_observation'].shape == (4096, 60)
            assert not torch.equal(info['final_observation'], obs)  # post-reset obs differs
        else:
            assert int(trunc.sum()) == 0 and 'final_observation' not in info
    x = torch.stack(all_obs).double()
    assert abs(float(x.mean())) < 5e-3 and abs(float(x.std()) - 1) < 5e-3
    assert abs(float(torch.stack(rewards).double().std()) - 1) < 2e-2
    assert abs(float(torch.stack(costs).mean()) - 0.05) < 5e-3
    assert not torch.equal(all_obs[0], all_obs[1])
    # determinism: same seed -> same stream
    env2 = envs.make('SynthPointGoal1-v0', num_envs=4096, device=DEV, horizon=5, cost_p=0.05)
    env2.set_seed(3)
    o2, _ = env2.reset()
    assert torch.equal(o2, all_obs[0])


def test_agent_ppolag_end_to_end(tmp_path):
    """omnisafe_amd.Agent('PPOLag', ...).learn(): two epochs on the synthetic env; csv columns are the
    reference's; checkpoint carries the reference's keys."""
    import omnisafe_amd

    cfg = {'seed': 1,
           'train_cfgs': {'device': DEV, 'total_steps': 2 * 64 * 40, 'vector_env_nums': 64},
           'algo_cfgs': {'steps_per_epoch': 64 * 40, 'update_iters': 2, 'batch_size': 64},
           'logger_cfgs': {'log_dir': str(tmp_path), 'save_model_freq': 1},
           'env_cfgs': {'horizon': 10, 'cost_p': 0.5}}
    agent = omnisafe_amd.Agent('PPOLag', 'SynthTiny-v0', custom_cfgs=cfg)
    ep_ret, ep_cost, ep_len = agent.learn()
    assert ep_len == 10.0 and 3.0 < ep_cost < 7.0 and abs(ep_ret) < 5.0
    rows = list(csv.DictReader(open(glob.glob(os.path.join(str(tmp_path), '*', '*', 'progress.csv'))[0])))
    assert len(rows) == 2
    expected = ['Metrics/EpRet', 'Metrics/EpCost', 'Metrics/EpLen', 'Train/Epoch', 'Train/Entropy',
                'Train/KL', 'Train/StopIter', 'Train/PolicyRatio', 'Train/PolicyRatio/Min',
                'Train/PolicyRatio/Max', 'Train/PolicyRatio/Std', 'Train/LR', 'Train/PolicyStd',
                'TotalEnvSteps', 'Loss/Loss_pi', 'Loss/Loss_pi/Delta', 'Value/Adv',
                'Loss/Loss_reward_critic', 'Loss/Loss_reward_critic/Delta', 'Value/reward',
                'Loss/Loss_cost_critic', 'Loss/Loss_cost_critic/Delta', 'Value/cost', 'Time/Total',
                'Time/Rollout', 'Time/Update', 'Time/Epoch', 'Time/FPS', 'Metrics/LagrangeMultiplier',
                'Metrics/LagrangeMultiplier/Min', 'Metrics/LagrangeMultiplier/Max',
                'Metrics/LagrangeMultiplier/Std']
    assert list(rows[0]) == expected  # reference column set and order (policy_gradient.py:133-236)
    assert float(rows[1]['TotalEnvSteps']) == 2 * 64 * 40 and float(rows[1]['Time/FPS']) > 0
    assert all(np.isfinite(float(v)) for v in rows[1].values())
    assert float(rows[1]['Train/LR']) == 0.0  # LinearLR reaches 0 after the last epoch
    ck = torch.load(glob.glob(os.path.join(str(tmp_path), '*', '*', 'torch_save', 'epoch-2.pt'))[0])
    assert list(ck['pi'])[0] == 'log_std' and ck['pi']['mean.0.weight'].shape == (64, 6)
    assert set(ck['obs_normalizer']) == {'_mean', '_sumsq', '_var', '_std', '_count', '_clip'}


def test_reward_and_cost_normalize_wrappers():
    """RewardNormalize / CostNormalize (envs/wrapper.py:280-423): the buffer receives the normalised
    streams (scalar running statistics over each step's batch, clip 5), episode metrics the originals."""
    from omnisafe_amd.adapter import OnPolicyAdapter
    from omnisafe_amd.buffer import VectorOnPolicyBuffer
    from test_mlp_gpu import make_ac

    N, T = 256, 12
    cfgs = _cfgs(reward_normalize=True, cost_normalize=True)
    cfgs.env_cfgs = types.SimpleNamespace(todict=lambda: {'horizon': 6, 'cost_p': 0.3})
    adapter = OnPolicyAdapter('SynthTiny-v0', N, 3, cfgs)
    ac = make_ac(6, 2)
    buf = VectorOnPolicyBuffer(adapter.observation_space, adapter.action_space, T, 0.99, 0.95, 0.95, 'gae',
                               0.0, True, True, num_envs=N, device=DEV)
    raw_r, raw_c = [], []
    env_step = adapter._env.step

    def spy(a):
        out = env_step(a)
        raw_r.append(out[1].clone())
        raw_c.append(out[2].clone())
        return out

    adapter._env.step = spy
    logger = _LoggerStub()
    adapter.rollout(T, ac, buf, logger)
    rn, cn = O.Normalizer((), clip=5), O.Normalizer((), clip=5)
    for t in range(T):
        np.testing.assert_allclose(buf.data['reward'][t].cpu().numpy(), rn.normalize(raw_r[t].cpu()).numpy(),
                                   rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(buf.data['cost'][t].cpu().numpy(), cn.normalize(raw_c[t].cpu()).numpy(),
                                   rtol=1e-4, atol=1e-5)
    # episode metrics use the original rewards / costs
    ep_cost = torch.stack(raw_c).cpu().reshape(2, 6, N).sum(1).reshape(-1)
    assert np.allclose(sorted(logger.data['Metrics/EpCost']), sorted(ep_cost.numpy()[-100:]), atol=1e-5) or \
        np.allclose(np.float32(logger.data['Metrics/EpCost']), ep_cost.numpy()[-100:], atol=1e-5)
    assert set(adapter.save()) == {'obs_normalizer', 'reward_normalizer', 'cost_normalizer'}


@pytest.mark.parametrize('tag', ['saute', 'simmer'])
def test_saute_simmer_rollout_on_reference_trace(golden, tag):
    """SauteAdapter / SimmerAdapter: replay the raw env outputs + policy noise recorded from the reference's
    PPOSaute / PPOSimmerPID rollout (tests/golden/{saute,simmer}_rollout.npz); the augmented observations
    (safety-state column), shaped rewards, bootstrap targets and Metrics/EpBudget must match."""
    from omnisafe_amd.adapter import SauteAdapter, SimmerAdapter
    from omnisafe_amd.buffer import VectorOnPolicyBuffer
    from test_mlp_gpu import make_ac

    g = golden(f'{tag}_rollout.npz')
    N, T = int(g['N']), int(g['T'])
    ns = types.SimpleNamespace
    extra = dict(safety_budget=1.0, saute_gamma=0.999, max_ep_len=16, unsafe_reward=-0.5)
    if tag == 'simmer':
        extra['upper_budget'] = 2.0
    cfgs = _cfgs(**extra)
    cfgs.control_cfgs = ns(kp=0.05, ki=0.01, kd=0.02, polyak=0.9)
    env = TraceEnv(g)
    adapter = (SimmerAdapter if tag == 'simmer' else SauteAdapter)('trace', N, 0, cfgs, env=env)
    assert adapter.observation_space.shape == (61,)
    ac = make_ac(61, 2, g, 'init/')
    eps_iter = iter(g['rollout/eps'])
    plain_step = ac.step

    def step_with_recorded_noise(obs, deterministic=False, eps=None, out=None, nets_mask=7):
        if out is not None and 'act' in out and not deterministic:
            eps = torch.from_numpy(next(eps_iter)).to(DEV)
        return plain_step(obs, deterministic=deterministic, eps=eps, out=out, nets_mask=nets_mask)

    ac.step = step_with_recorded_noise
    buf = VectorOnPolicyBuffer(adapter.observation_space, adapter.action_space, T, 0.99, 0.95, 0.95,
                               'gae', 0.0, True, True, num_envs=N, device=DEV)
    logger = _LoggerStub()
    adapter.rollout(T, ac, buf, logger)
    buf.compute_advantages()
    b = {k: v.cpu().numpy() for k, v in buf.data.items()}
    # the safety-state column is float32 arithmetic restated op for op: bit-exact
    assert np.array_equal(b['obs'][:, :, 60], g['buffer/obs'][:, :, 60])
    assert np.array_equal(b['reward'], g['buffer/reward']) and np.array_equal(b['cost'], g['buffer/cost'])
    assert (b['reward'] == np.float32(-0.5)).sum() > 20  # the unsafe branch is exercised
    for k in ('obs', 'act', 'value_r', 'value_c', 'logp'):
        np.testing.assert_allclose(b[k], g[f'buffer/{k}'], rtol=1e-4, atol=2e-5, err_msg=k)
    for k in ('adv_r', 'adv_c', 'target_value_r', 'target_value_c'):
        np.testing.assert_allclose(b[k], g[f'buffer/{k}'], rtol=1e-4, atol=1e-4, err_msg=k)
    np.testing.assert_allclose(logger.data['Metrics/EpBudget'], g['rollout/ep_budget_window'], rtol=1e-6)
    np.testing.assert_allclose(logger.data['Metrics/EpRet'], g['rollout/ep_ret_window'], rtol=1e-6)
    assert np.array_equal(np.float32(logger.data['Metrics/EpCost']), g['rollout/ep_cost_window'])
    if tag == 'simmer':  # PID budget controller trajectory (simmer_agent.py:131-176)
        for jc, want in zip(g['control/jc'], g['control/budget_rel']):
            adapter.control_budget(float(jc))
            got = np.array([float(adapter._safety_budget[0]), float(adapter._rel_budget[0])], np.float32)
            np.testing.assert_allclose(got, want, rtol=1e-6)


@pytest.mark.parametrize('algo_name', ['PPOSaute', 'TRPOSaute', 'PPOSimmerPID', 'TRPOSimmerPID'])
def test_saute_simmer_agents_end_to_end(tmp_path, algo_name):
    import omnisafe_amd

    cfg = {'seed': 2, 'train_cfgs': {'device': DEV, 'total_steps': 2 * 128 * 32, 'vector_env_nums': 128},
           'algo_cfgs': {'steps_per_epoch': 128 * 32, 'update_iters': 2, 'safety_budget': 2.0, 'max_ep_len': 16},
           'logger_cfgs': {'log_dir': str(tmp_path), 'verbose': False}, 'env_cfgs': {'horizon': 16, 'cost_p': 0.3}}
    agent = omnisafe_amd.Agent(algo_name, 'SynthCarGoal1-v0', custom_cfgs=cfg)
    assert agent.agent._actor_critic.obs_dim == 73  # CarGoal1's 72 + the safety state
    ep_ret, ep_cost, ep_len = agent.learn()
    assert ep_len == 16.0 and 2.0 < ep_cost < 8.0 and np.isfinite(ep_ret)
    assert len(agent.agent._logger._data) > 0


@pytest.mark.parametrize('algo_name,env_id,obs_dim', [('CPO', 'SynthCarGoal1-v0', 72), ('PPOLag', 'SynthHumanoid-v0', 376),
                                                      ('TRPOLag', 'SynthAnt-v0', 27)])
def test_baseline_config_shapes_end_to_end(tmp_path, algo_name, env_id, obs_dim):
    """BASELINE.json configs 3-5 on their observation / action shapes (one GPU): CarGoal1 72/2, Humanoid 376/17
    (wider than the persistent kernel supports: per-step kernels), Ant 27/8 (unaligned rows: padded once per
    update).  Two epochs through `Agent.learn()`; finite parameters, sane episode statistics."""
    import omnisafe_amd

    cfg = {'seed': 1, 'train_cfgs': {'device': DEV, 'total_steps': 2 * 64 * 32, 'vector_env_nums': 64},
           'algo_cfgs': {'steps_per_epoch': 64 * 32, 'update_iters': 2},
           'logger_cfgs': {'log_dir': str(tmp_path), 'verbose': False}, 'env_cfgs': {'horizon': 16, 'cost_p': 0.3}}
    agent = omnisafe_amd.Agent(algo_name, env_id, custom_cfgs=cfg)
    ac = agent.agent._actor_critic
    assert ac.obs_dim == obs_dim
    before = ac.params.clone()
    ep_ret, ep_cost, ep_len = agent.learn()
    assert ep_len == 16.0 and 2.0 < ep_cost < 8.0 and np.isfinite(ep_ret)
    assert bool(torch.isfinite(ac.params).all()) and not torch.equal(before, ac.params)


class _TraceEnvWithResets(TraceEnv):
    """TraceEnv whose reset() serves the reference's recorded resets in call order (the early-terminated
    adapter resets the env in the middle of an epoch) and does not rewind the trace."""

    def __init__(self, g, dev=None):
        super().__init__(g, dev=dev)
        self.n_resets = 0

    def reset(self, seed=None, options=None):
        obs = torch.from_numpy(self.g['rollout/resets'][self.n_resets]).to(self.dev)
        self.n_resets += 1
        return obs, {}


@pytest.mark.parametrize('host_env', [False, True], ids=['device-env', 'host-env-bridge'])
def test_early_terminated_rollout_on_reference_trace(golden, host_env):
    """EarlyTerminatedAdapter (early_terminated_adapter.py:50-88) replayed on the raw env trace of a
    reference PPOEarlyTerminated rollout: 14 early terminations (zero reward, terminated, mid-epoch
    reset, no bootstrap) interleaved with 18 time-limit truncations.  `host-env-bridge`: the same trace served by a
    HOST env through HostEnvBridge UNDER the early-termination wrapper -- the bootstrap values / episode accounting of
    a step are handed to the bridge (`defer`, reached through the wrapper's attribute forwarding) and must still be
    written for every step (round-4 advisor finding: an attribute assignment landed on the wrapper and was lost)."""
    from omnisafe_amd.adapter import EarlyTerminatedAdapter, HostEnvBridge
    from omnisafe_amd.buffer import VectorOnPolicyBuffer
    from test_mlp_gpu import make_ac

    g = golden('early_terminated_rollout.npz')
    N, T = int(g['N']), int(g['T'])
    env = _TraceEnvWithResets(g, dev='cpu' if host_env else None)
    bridge = HostEnvBridge(env, DEV) if host_env else None
    adapter = EarlyTerminatedAdapter('trace', N, 0, _cfgs(cost_limit=float(g['cost_limit'])),
                                     env=bridge if host_env else env)
    if host_env:
        deferred = []
        plain_defer = bridge.defer
        bridge.defer = lambda work: (deferred.append(1), plain_defer(work))[1]
    ac = make_ac(60, 2, g, 'init/')
    eps_iter = iter(g['rollout/eps'])
    plain_step = ac.step

    def step_with_recorded_noise(obs, deterministic=False, eps=None, out=None, nets_mask=7):
        if out is not None and 'act' in out and not deterministic:
            eps = torch.from_numpy(next(eps_iter)).to(DEV)
        return plain_step(obs, deterministic=deterministic, eps=eps, out=out, nets_mask=nets_mask)

    ac.step = step_with_recorded_noise
    buf = VectorOnPolicyBuffer(adapter.observation_space, adapter.action_space, T, 0.99, 0.95, 0.95,
                               'gae', 0.0, True, True, num_envs=N, device=DEV)
    logger = _LoggerStub()
    adapter.rollout(T, ac, buf, logger)
    assert env.n_resets == g['rollout/resets'].shape[0] == 15  # epoch start + 14 early terminations
    if host_env:  # every step but the epoch's last handed its post-step work to the bridge, and none is left over
        assert len(deferred) == T - 1 and bridge.deferred_device_work is None
    buf.compute_advantages()
    b = {k: v.cpu().numpy() for k, v in buf.data.items()}
    assert np.array_equal(b['reward'], g['buffer/reward']) and np.array_equal(b['cost'], g['buffer/cost'])
    # early-terminated steps: reward zeroed although the env returned a non-zero one
    early = (b['reward'][:, 0] == 0) & (g['rollout/reward'][:, 0] != 0)
    assert early.sum() == 14
    pe = (g['rollout/truncated'][:, 0].astype(bool) | early).astype(np.uint8)
    pe[-1] = 1
    assert np.array_equal(b['path_end'][:, 0], pe)
    assert not b['boot_r'][early, 0].any() and not b['boot_c'][early, 0].any()  # terminated: no bootstrap
    for k in ('obs', 'act', 'value_r', 'value_c', 'logp'):
        np.testing.assert_allclose(b[k], g[f'buffer/{k}'], rtol=1e-4, atol=2e-5, err_msg=k)
    for k in ('adv_r', 'adv_c', 'target_value_r', 'target_value_c'):
        np.testing.assert_allclose(b[k], g[f'buffer/{k}'], rtol=1e-4, atol=1e-4, err_msg=k)
    norm = adapter.save()['obs_normalizer']
    assert int(norm._count) == int(g['rollout/norm_count']) == 1 + T + 14 + 18
    np.testing.assert_allclose(norm.mean.cpu().numpy(), g['rollout/norm_mean'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(norm.std.cpu().numpy(), g['rollout/norm_std'], rtol=1e-5)
    np.testing.assert_allclose(logger.data['Metrics/EpRet'], g['rollout/ep_ret_window'], rtol=1e-6, atol=1e-6)
    assert np.array_equal(np.float32(logger.data['Metrics/EpCost']), g['rollout/ep_cost_window'])
    assert np.array_equal(np.float32(logger.data['Metrics/EpLen']), g['rollout/ep_len_window'])
    with pytest.raises(AssertionError, match='only supports num_envs=1'):
        EarlyTerminatedAdapter('trace', 2, 0, _cfgs(cost_limit=1.0), env=env)


@pytest.mark.parametrize('algo_name', ['PPOEarlyTerminated', 'TRPOEarlyTerminated'])
def test_early_terminated_agents_end_to_end(tmp_path, algo_name):
    import omnisafe_amd

    cfg = {'seed': 4, 'train_cfgs': {'device': DEV, 'total_steps': 2 * 512, 'vector_env_nums': 1},
           'algo_cfgs': {'steps_per_epoch': 512, 'update_iters': 2, 'cost_limit': 2.5},
           'logger_cfgs': {'log_dir': str(tmp_path), 'verbose': False}}
    agent = omnisafe_amd.Agent(algo_name, 'SynthTiny-v0', custom_cfgs=cfg)
    ep_ret, ep_cost, ep_len = agent.learn()
    # cost arrives in units of 1 with p = 0.05 per step: an episode ends when the accumulator reaches 3
    # (it may start above zero: it survives the epoch's reset); the 1000-step time limit is never reached
    assert np.isfinite(ep_ret) and 0 < ep_cost <= 3.0 and ep_len < 512


@pytest.mark.parametrize('shape', [(), (10,), (10, 10)])
def test_normalizer_like_the_reference_test(shape):
    """The reference's own test of this class (tests/test_normalizer.py:24-52): 1000 single samples, then
    1000 batches of 10, of standard normal data; running mean / std within 1e-2 of the sample statistics
    (here also within 1e-5 of them: the running update is exact up to float32 rounding)."""
    from omnisafe_amd.normalizer import Normalizer

    gen = torch.Generator(device='cpu').manual_seed(len(shape))
    norm = Normalizer(shape, device=DEV)
    assert norm.mean.shape == shape
    data_lst = []
    for _ in range(1000):
        data = torch.randn(shape, generator=gen)
        data_lst.append(data)
        out = norm(data.to(DEV))
        assert out.shape == shape
    data = torch.stack(data_lst)
    assert torch.allclose(data.mean(dim=0), norm.mean.cpu(), atol=1e-2)
    assert torch.allclose(data.std(dim=0), norm.std.cpu(), atol=1e-2)
    assert torch.allclose(data.mean(dim=0), norm.mean.cpu(), atol=1e-5)
    assert torch.allclose(data.std(dim=0), norm.std.cpu(), rtol=1e-4)
    norm = Normalizer(shape, device=DEV)
    data_lst = []
    for _ in range(1000):
        data = torch.randn(10, *shape, generator=gen)
        data_lst.append(data)
        norm(data.to(DEV))
    data = torch.cat(data_lst)
    assert torch.allclose(data.mean(dim=0), norm.mean.cpu(), atol=1e-2)
    assert torch.allclose(data.std(dim=0), norm.std.cpu(), atol=1e-2)
    assert torch.allclose(data.std(dim=0), norm.std.cpu(), rtol=1e-4)
    assert int(norm._count) == 10000 and norm.state_dict()['_mean'].shape == shape


def test_exploration_noise_anneal_end_to_end(tmp_path):
    """model_cfgs.exploration_noise_anneal (policy_gradient.py:101-105, 271-272): after every update the
    actor's std is reset to the schedule's value for that epoch."""
    import omnisafe_amd

    cfg = {'seed': 3, 'train_cfgs': {'device': DEV, 'total_steps': 3 * 64 * 20, 'vector_env_nums': 64},
           'algo_cfgs': {'steps_per_epoch': 64 * 20, 'update_iters': 2},
           'model_cfgs': {'exploration_noise_anneal': True, 'std_range': [0.5, 0.1]},
           'logger_cfgs': {'log_dir': str(tmp_path), 'verbose': False}, 'env_cfgs': {'horizon': 10}}
    agent = omnisafe_amd.Agent('PPOLag', 'SynthTiny-v0', custom_cfgs=cfg)
    agent.learn()
    assert agent.agent._actor_critic.actor.std == pytest.approx(0.5 + 2 / 3 * (0.1 - 0.5), rel=1e-6)
    rows = list(csv.DictReader(open(glob.glob(os.path.join(str(tmp_path), '*', '*', 'progress.csv'))[0])))
    # Train/PolicyStd of epoch e is logged during its update, i.e. before that epoch's annealing: epoch 1
    # starts from the value set after epoch 0 (0.5) and epoch 2 from 0.5 - 0.4/3, each moved a little by Adam
    assert abs(float(rows[1]['Train/PolicyStd']) - 0.5) < 0.02
    assert abs(float(rows[2]['Train/PolicyStd']) - (0.5 - 0.4 / 3)) < 0.02


@pytest.mark.parametrize('algo_name,env_id', [('PPOLag', 'SynthPointGoal1-v0'), ('PPOSaute', 'SynthReach-v0'),
                                              ('TRPOLag', 'SynthAnt-v0')])
def test_rollout_graph_replay_equals_eager_launches(tmp_path, monkeypatch, algo_name, env_id):
    """The hipGraph of an epoch's rollout (captured on the second epoch, replayed afterwards) against the same
    agent run with eager launches: bit-identical buffers, normaliser state, episode metrics and parameters
    after every one of 5 epochs (rollout + update), i.e. the device-resident Philox stream positions advance
    exactly as the host counters do, and truncation steps (final observations, bootstrap values) replay."""
    import omnisafe_amd

    def run(graph):
        monkeypatch.setenv('OSA_ROLLOUT_GRAPH', '1' if graph else '0')
        cfg = {'seed': 7, 'train_cfgs': {'device': DEV, 'total_steps': 5 * 64 * 24, 'vector_env_nums': 64},
               'algo_cfgs': {'steps_per_epoch': 64 * 24, 'update_iters': 2},
               'logger_cfgs': {'log_dir': str(tmp_path / ('g' if graph else 'e')), 'verbose': False}}
        if env_id != 'SynthReach-v0':
            cfg['env_cfgs'] = {'horizon': 10, 'cost_p': 0.2}  # truncations at steps 10 and 20 of the 24
        algo = omnisafe_amd.Agent(algo_name, env_id, custom_cfgs=cfg).agent
        snaps = []
        for _ in range(5):
            algo._env.rollout(steps_per_epoch=algo._steps_per_epoch, agent=algo._actor_critic, buffer=algo._buf,
                              logger=algo._logger)
            snap = {k: v.clone() for k, v in algo._buf.data.items()}
            snap['norm_mean'] = algo._env._obs_normalizer._mean.clone()
            snap['ep_cost'] = torch.tensor(list(algo._logger._data['Metrics/EpCost']))
            algo._update()
            snap['params'] = algo._actor_critic.params.clone()
            snaps.append(snap)
            algo._logger.dump_tabular()
        return algo, snaps

    a_g, s_g = run(True)
    a_e, s_e = run(False)
    assert a_g._env.last_rollout_graphed is True and not getattr(a_e._env, 'last_rollout_graphed', False)
    for ep, (g, e) in enumerate(zip(s_g, s_e)):
        for k in g:
            assert torch.equal(g[k].cpu(), e[k].cpu()), (ep, k)
    # epochs differ from each other (fresh noise every epoch, not a replay of the captured numbers)
    assert not torch.equal(s_g[2]['act'], s_g[3]['act']) and not torch.equal(s_g[3]['reward'], s_g[4]['reward'])


class _CountEnv:
    """Single, NON-auto-resetting env that asks for the TimeLimit and AutoReset wrappers (the shape of a
    caller-side Safety-Gymnasium shim, envs/safety_gymnasium_env.py:160-210, single-env case): obs = [steps since
    reset, episode index, 0, 0], reward 1, cost 0.5, terminates by itself after 5 steps in every third episode."""
    need_auto_reset_wrapper = True
    need_time_limit_wrapper = True
    need_evaluation = False
    num_envs = 1
    max_episode_steps = 7

    def __init__(self):
        from omnisafe_amd.spaces import Box

        self.observation_space = Box(-np.inf, np.inf, (4,))
        self.action_space = Box(-2.0, 2.0, (2,))
        self.k, self.ep, self.resets = 0, -1, 0

    def set_seed(self, seed):
        pass

    def _obs(self):
        return torch.tensor([float(self.k), float(self.ep), 0.0, 0.0], device=DEV)

    def reset(self, seed=None, options=None):
        self.k, self.ep, self.resets = 0, self.ep + 1, self.resets + 1
        return self._obs(), {}

    def step(self, action):
        assert action.shape[-1] == 2
        self.k += 1
        term = self.ep % 3 == 2 and self.k >= 5
        return (self._obs(), torch.tensor(1.0, device=DEV), torch.tensor(0.5, device=DEV),
                torch.tensor(term, device=DEV), torch.tensor(False, device=DEV), {})

    def close(self):
        pass


def test_time_limit_and_auto_reset_wrappers():
    """online_adapter.py:120-132 + wrapper.py:31-176 for a single env that needs both wrappers: episodes are cut
    after max_episode_steps = 7 (truncation: bootstrap with V(final observation)) or end by themselves (termination:
    bootstrap 0); the observation after an episode end is the first of the new episode."""
    from omnisafe_amd.adapter import AutoReset, OnPolicyAdapter, TimeLimit
    from omnisafe_amd.buffer import VectorOnPolicyBuffer
    from test_mlp_gpu import make_ac

    env = _CountEnv()
    adapter = OnPolicyAdapter('count', 1, 0, _cfgs(obs_normalize=False), env=env)
    assert isinstance(adapter._env, AutoReset) and isinstance(adapter._env._env, TimeLimit)
    T = 24
    torch.manual_seed(2)
    ac = make_ac(4, 2)
    buf = VectorOnPolicyBuffer(adapter.observation_space, adapter.action_space, T, 0.99, 0.95, 0.95, 'gae', 0.0, True,
                               True, num_envs=1, device=DEV)
    logger = _LoggerStub()
    adapter.rollout(T, ac, buf, logger)
    b = {k: v.cpu().numpy() for k, v in buf.data.items()}
    # episodes: 0 (7 steps, truncated), 1 (7, truncated), 2 (5, terminated), 3 (5 of 7 steps when the epoch ends)
    ends = [6, 13, 18, 23]
    assert b['path_end'][:, 0].nonzero()[0].tolist() == ends
    assert logger.data['Metrics/EpLen'] == [7.0, 7.0, 5.0] and logger.data['Metrics/EpRet'] == [7.0, 7.0, 5.0]
    assert logger.data['Metrics/EpCost'] == [3.5, 3.5, 2.5]
    want_obs = [[k, 0] for k in range(7)] + [[k, 1] for k in range(7)] + [[k, 2] for k in range(5)] + \
               [[k, 3] for k in range(5)]
    np.testing.assert_array_equal(b['obs'][:, 0, :2], np.asarray(want_obs, np.float32))
    # bootstraps: V(final observation) at truncations, 0 at the termination, V(next observation) at the epoch end
    v = lambda o: ac.values(torch.tensor([o], dtype=torch.float32, device=DEV))  # noqa: E731
    for t, o in ((6, [7.0, 0.0, 0.0, 0.0]), (13, [7.0, 1.0, 0.0, 0.0]), (23, [5.0, 3.0, 0.0, 0.0])):
        vr, vc = v(o)
        np.testing.assert_allclose(b['boot_r'][t, 0], float(vr[0]), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(b['boot_c'][t, 0], float(vc[0]), rtol=1e-5, atol=1e-6)
    assert b['boot_r'][18, 0] == 0.0 and b['boot_c'][18, 0] == 0.0
    assert env.resets == 1 + 3  # the epoch's reset + one per finished episode
    # a vector env asking for the wrappers is refused like in the reference (single env only)
    env2 = _CountEnv()
    env2.num_envs = 2
    with pytest.raises(AssertionError, match='single environment'):
        OnPolicyAdapter('count', 2, 0, _cfgs(obs_normalize=False), env=env2)


@pytest.mark.gpu
def test_policy_step_with_fused_action_scale_equals_the_two_launches():
    """osa_policy_step_scaled: ActionScale.step (envs/wrapper.py:510-514) in the policy step's launch -- the same
    actions, values and log-probabilities as osa_policy_step, and the bits of osa_action_scale on those actions."""
    import types

    from omnisafe_amd import _lib
    from omnisafe_amd.models import ConstraintActorCritic
    from omnisafe_amd.spaces import Box

    ns = types.SimpleNamespace
    mc = ns(actor=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4),
            critic=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4),
            weight_initialization_mode='kaiming_uniform', actor_type='gaussian_learning', linear_lr_decay=True)
    lib = _lib.load(require_gpu=True)
    for d_o, d_a, N in ((60, 2, 4096), (27, 8, 1000), (72, 17, 130)):
        torch.manual_seed(d_o)
        ac = ConstraintActorCritic(Box(-np.inf, np.inf, (d_o,)), Box(-1, 1, (d_a,)), mc, 4, device=DEV)
        ac.set_seed(5)
        ac._rng_offset = 0  # (the stream position advances with every step: same position for both calls)
        obs = torch.randn(N, d_o, device=DEV)
        lo = torch.linspace(-2.0, -0.5, d_a, device=DEV)
        hi = torch.linspace(0.7, 3.0, d_a, device=DEV)
        env_a = torch.full((N, d_a), 7.0, device=DEV)
        a1, vr1, vc1, lp1 = ac.step(obs, out={'scale': (env_a, lo, hi, -1.0, 1.0)})
        ac._rng_offset = 0
        a2, vr2, vc2, lp2 = ac.step(obs)
        env_b = torch.empty_like(env_a)
        _lib.check(lib.osa_action_scale(_lib.ptr(a2), d_a, _lib.ptr(env_b), d_a, N, d_a, _lib.ptr(lo), _lib.ptr(hi),
                                        -1.0, 1.0, _lib.stream_ptr()), 'osa_action_scale')
        for x, y in ((a1, a2), (vr1, vr2), (vc1, vc2), (lp1, lp2), (env_a, env_b)):
            assert torch.equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize('T,N,p', [(16, 4096, 0.06), (3, 5, 0.5), (1, 1, 1.0), (50, 100, 0.0), (300, 4096, 0.01),
                                   (64, 16411, 0.2)])
def test_episode_flush_kernel_equals_the_torch_form(T, N, p):
    """osa_episode_flush (two launches) vs what rounds 1-3 did with torch (nonzero, gathers, means): the finished
    episodes in (step, env) order with their return / cost / length / extra column, their count, and the means of the
    two value columns -- over several grid sizes, a ragged last range, no finished episode at all, and twice on ONE
    workspace with different sizes (the ticket has a fixed place)."""
    from omnisafe_amd import _lib

    lib = _lib.load(require_gpu=True)
    ws = torch.zeros(lib.osa_episode_flush_ws_doubles(1 << 22), dtype=torch.float64, device=DEV)
    for rep, (t, n) in enumerate(((T, N), (max(1, T // 2), N), (T, N))):
        M = t * n
        g = torch.Generator(device=DEV).manual_seed(7 + rep)
        done = (torch.rand(M, device=DEV, generator=g) < p).to(torch.uint8)
        ret, cost, ln, extra, vr, vc = (torch.randn(M, device=DEV, generator=g) for _ in range(6))
        hdr = torch.zeros(4, dtype=torch.int32, device=DEV)
        idx = torch.full((M,), -1, dtype=torch.int32, device=DEV)
        vals = torch.full((4 * M,), 7.0, device=DEV)
        _lib.check(lib.osa_episode_flush(_lib.ptr(done), _lib.ptr(ret), _lib.ptr(cost), _lib.ptr(ln), _lib.ptr(extra), M,
                                         _lib.ptr(vr), _lib.ptr(vc), _lib.ptr(hdr[0:1]), _lib.ptr(idx), _lib.ptr(vals),
                                         _lib.ptr(hdr[1:3]), _lib.ptr(ws), _lib.stream_ptr()), 'osa_episode_flush')
        want = done.nonzero().reshape(-1)
        cnt = int(hdr[0])
        assert cnt == want.numel()
        assert torch.equal(idx[:cnt].long(), want) and bool((idx[cnt:] == -1).all())
        v = vals.view(4, M)
        for k, src in enumerate((ret, cost, ln, extra)):
            assert torch.equal(v[k, :cnt], src[want])
        means = hdr[1:3].view(torch.float32).cpu().numpy()
        np.testing.assert_allclose(means, [float(vr.double().mean()), float(vc.double().mean())], rtol=1e-6, atol=1e-7)
        assert int(ws.view(torch.int32)[0]) == 0  # ticket re-armed
    out = torch.empty(1, device=DEV)
    x = torch.randn(100000, device=DEV)
    ii = torch.randint(0, 100000, (16384,), device=DEV)
    _lib.check(lib.osa_gather_mean(_lib.ptr(x), _lib.ptr(ii), ii.numel(), _lib.ptr(out), _lib.stream_ptr()), 'osa_gather_mean')
    np.testing.assert_allclose(float(out), float(x[ii].double().mean()), rtol=1e-6, atol=1e-8)
    _lib.check(lib.osa_gather_mean(_lib.ptr(x), None, 77, _lib.ptr(out), _lib.stream_ptr()), 'osa_gather_mean')
    np.testing.assert_allclose(float(out), float(x[:77].double().mean()), rtol=1e-6, atol=1e-8)
