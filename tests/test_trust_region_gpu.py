"""GPU parity of the natural-gradient path (TRPOLag / CPO actor update) against golden vectors recorded
from the unmodified reference (oracle/make_golden.py::gen_trust_region_updates).

Tolerances: policy gradient rtol 1e-3 / atol 2e-6; Fisher-vector product rtol 2e-3
(JVP->VJP vs the reference's autograd double backward); CG solution rtol 2e-2 of its norm (15
iterations amplify float32 noise); accepted line-search index and CPO case identical; post-update
parameters atol 2e-4 (they are theta_old + a step of norm ~0.3)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _algo(name, tmp_path, g):
    import omnisafe_amd

    N, T = int(g['N']), int(g['T'])
    cfg = {'seed': 0, 'train_cfgs': {'device': DEV, 'total_steps': 4 * N * T, 'vector_env_nums': N},
           'algo_cfgs': {'steps_per_epoch': N * T, 'update_iters': 2, 'batch_size': 128},
           'logger_cfgs': {'log_dir': str(tmp_path), 'verbose': False}, 'env_cfgs': {'horizon': 16}}
    algo = omnisafe_amd.Agent(name, 'SynthPointGoal1-v0', custom_cfgs=cfg).agent
    ac = algo._actor_critic
    for net in ('actor', 'reward_critic', 'cost_critic'):
        sd = {k[len('init/') + len(net) + 1:]: torch.from_numpy(v.copy()) for k, v in g.items()
              if k.startswith(f'init/{net}/')}
        getattr(ac, net).load_state_dict(sd)
    data = {k[5:]: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in g.items() if k.startswith('data/')}
    return algo, ac, data


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def test_trpolag_actor_update_vs_reference(golden, tmp_path):
    g = golden('trpolag_actor_update.npz')
    algo, ac, data = _algo('TRPOLag', tmp_path, g)
    algo._lagrange._param.data.fill_(float(g['lambda']))
    algo._lagrange._device_copy.fill_(float(g['lambda']))
    s = algo._solver
    # ---- pieces
    loss, gr = algo._policy_gradient(data)
    gflat = ac.actor.unpad(gr).cpu().numpy()
    np.testing.assert_allclose(gflat, g['g'], rtol=1e-3, atol=2e-6)
    np.testing.assert_allclose(float(loss), g['loss_before'], atol=1e-6)
    xg = ac.actor.pad(torch.from_numpy(g['x']))
    Fx = ac.actor.unpad(s.fvp(xg)).cpu().numpy()
    assert _rel(Fx, g['Fx']) < 2e-3
    x = ac.actor.unpad(s.conjugate_gradients(gr)).cpu().numpy()
    assert _rel(x, g['x']) < 2e-2
    # ---- whole update
    algo._update_actor(data)
    info = algo._last_actor_update
    assert info['accept_step'] == int(g['accept_step'])
    np.testing.assert_allclose(info['xHx'], g['log/Misc/xHx'][0], rtol=1e-2)
    np.testing.assert_allclose(info['alpha'], g['log/Misc/Alpha'][0], rtol=1e-2)
    step = ac.actor.unpad(info['final_step']).cpu().numpy()
    assert _rel(step, g['final_step']) < 2e-2
    for k, v in ac.actor.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g[f'post/actor/{k}'], rtol=0, atol=3e-4, err_msg=k)
    # padding of every flat vector stays zero
    lay = ac.layout
    mask = torch.ones(lay.P, dtype=torch.bool, device=DEV)
    mask[ac.actor._flat_index] = False
    assert float(ac.params[0][mask].abs().max()) == 0.0
    assert float(info['final_step'][mask].abs().max()) == 0.0


def test_cpo_actor_update_vs_reference(golden, tmp_path):
    g = golden('cpo_actor_update.npz')
    algo, ac, data = _algo('CPO', tmp_path, g)
    algo._logger.extend('Metrics/EpCost', [float(g['ep_cost_mean'])])
    algo._update_actor(data)
    info = algo._last_actor_update
    np.testing.assert_allclose(ac.actor.unpad(info['g']).cpu().numpy(), g['g'], rtol=1e-3, atol=2e-6)
    np.testing.assert_allclose(ac.actor.unpad(info['b']).cpu().numpy(), g['b'], rtol=1e-3, atol=2e-6)
    assert _rel(ac.actor.unpad(info['x']).cpu().numpy(), g['x']) < 2e-2
    assert _rel(ac.actor.unpad(info['p']).cpu().numpy(), g['p']) < 2e-2
    assert info['case'] == int(g['optim_case'])
    np.testing.assert_allclose(info['q'], g['log/Misc/q'][0], rtol=1e-2)
    np.testing.assert_allclose(info['r'], g['log/Misc/r'][0], rtol=5e-2, atol=1e-4)
    np.testing.assert_allclose(info['s'], g['log/Misc/s'][0], rtol=1e-2)
    np.testing.assert_allclose(info['lambda_star'], g['log/Misc/Lambda_star'][0], rtol=1e-2)
    np.testing.assert_allclose(info['nu_star'], g['log/Misc/Nu_star'][0], atol=1e-6)
    np.testing.assert_allclose(info['loss_reward_before'], g['loss_reward_before'], atol=1e-6)
    np.testing.assert_allclose(info['loss_cost_before'], g['loss_cost_before'], atol=1e-6)
    assert info['accept_step'] == int(g['accept_step'])
    assert _rel(ac.actor.unpad(info['final_step']).cpu().numpy(), g['final_step']) < 2e-2
    for k, v in ac.actor.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g[f'post/actor/{k}'], rtol=0, atol=3e-4, err_msg=k)


def test_fvp_is_symmetric_positive_and_matches_oracle_autograd():
    """Size-independent properties of the Fisher-vector product at a BASELINE shape (CarGoal1 72/2,
    M = 8192): symmetry u.Fv = v.Fu, positivity v.Fv > 0, linearity; plus agreement with the oracle's
    double-backward FVP on a subsample."""
    import np_oracle as O
    from omnisafe_amd.trust_region import TrustRegionSolver
    from test_mlp_gpu import make_ac

    torch.manual_seed(5)
    ac = make_ac(72, 2)
    with torch.no_grad():
        ac.params[0, ac.layout.oLS:ac.layout.oLS + 2] = torch.tensor([-0.3, 0.2], device=DEV)
    obs = torch.randn(8192, 72, device=DEV)
    s = TrustRegionSolver(ac, cg_iters=15, cg_damping=0.1)
    s.begin(obs)
    u = ac.actor.pad(torch.randn(ac.actor.num_params))
    v = ac.actor.pad(torch.randn(ac.actor.num_params))
    Fu, Fv = s.fvp(u).clone(), s.fvp(v).clone()
    uFv, vFu, vFv = float(s.dot(u, Fv)), float(s.dot(v, Fu)), float(s.dot(v, Fv))
    assert abs(uFv - vFu) < 1e-3 * max(abs(uFv), 1e-3) and vFv > 0
    F2 = s.fvp(s.lincomb(2.0, u, -3.0, v))
    np.testing.assert_allclose(F2.cpu().numpy(), (2 * Fu - 3 * Fv).cpu().numpy(), rtol=2e-3, atol=2e-5)
    ref = O.Actor(72, 2)
    ref.load_state_dict({k: t.cpu() for k, t in ac.actor.state_dict().items()})
    Fv_ref = O.fvp(ref, obs[:512].cpu(), ac.actor.unpad(v).cpu(), cg_damping=0.1)
    s2 = TrustRegionSolver(ac, cg_iters=15, cg_damping=0.1)
    s2.begin(obs[:512])
    assert _rel(ac.actor.unpad(s2.fvp(v)).cpu().numpy(), Fv_ref.numpy()) < 2e-3
    # CG actually solves (F + damping) x = b
    b = ac.actor.pad(torch.randn(ac.actor.num_params) * 0.1)
    x = s.conjugate_gradients(b)
    res = (s.fvp(x) - b).norm() / b.norm()
    assert float(res) < 0.2  # 15 iterations on an 8964-dim system: residual clearly reduced


@pytest.mark.parametrize('obs_dim,act_dim,M', [(60, 2, 65536), (27, 8, 4096 + 37), (6, 2, 200), (64, 16, 1000),
                                               (60, 2, 64 * 300 + 1), (72, 2, 8192 + 5), (80, 3, 300)])
def test_fast_fvp_is_bit_identical_to_the_general_kernel(monkeypatch, obs_dim, act_dim, M):
    """The throughput-shaped Fisher-vector product (csrc/fvp_kernel.hip: theta and v in LDS, gradient in registers, one
    slab per workgroup) against the general gradient kernel it replaces for hidden width 64 / observations up to 64
    wide: the same MFMA sequences and summation orders, hence torch.equal -- for full and ragged last chunks, fewer
    chunks than workgroups, several chunks per workgroup, act_dim 2 ... 16; and through CG the same solution."""
    from omnisafe_amd.trust_region import TrustRegionSolver
    from test_mlp_gpu import make_ac

    torch.manual_seed(9)
    ac = make_ac(obs_dim, act_dim)
    with torch.no_grad():
        ac.params[0, ac.layout.oLS:ac.layout.oLS + act_dim] = torch.linspace(-0.4, 0.3, act_dim, device=DEV)
    obs = torch.randn(M, obs_dim, device=DEV)
    v = ac.actor.pad(torch.randn(ac.actor.num_params))
    b = ac.actor.pad(torch.randn(ac.actor.num_params) * 0.1)
    out = {}
    for fast in ('1', '0'):
        monkeypatch.setenv('OSA_FVP_FAST', fast)
        s = TrustRegionSolver(ac, cg_iters=10, cg_damping=0.1)
        s.begin(obs)
        out[fast] = (s.fvp(v).clone(), s.conjugate_gradients(b).clone())
    assert torch.equal(out['1'][0], out['0'][0]) and torch.equal(out['1'][1], out['0'][1])
    assert float(out['1'][0].abs().max()) > 0


@pytest.mark.parametrize('algo_name', ['TRPOLag', 'CPO', 'TRPO', 'NaturalPG', 'PPO', 'PolicyGradient'])
def test_agents_end_to_end(tmp_path, algo_name):
    import omnisafe_amd

    cfg = {'seed': 2, 'train_cfgs': {'device': DEV, 'total_steps': 2 * 128 * 32, 'vector_env_nums': 128},
           'algo_cfgs': {'steps_per_epoch': 128 * 32, 'update_iters': 2},
           'logger_cfgs': {'log_dir': str(tmp_path), 'verbose': False}, 'env_cfgs': {'horizon': 16, 'cost_p': 0.3}}
    ep_ret, ep_cost, ep_len = omnisafe_amd.Agent(algo_name, 'SynthCarGoal1-v0', custom_cfgs=cfg).learn()
    assert ep_len == 16.0 and 2.0 < ep_cost < 8.0 and np.isfinite(ep_ret)


@pytest.mark.parametrize('algo_name,width', [('PPOLag', 64), ('CPO', 64), ('PPOLag', 256), ('TRPOLag', 128)])
def test_agents_with_relu_networks(tmp_path, algo_name, width):
    """model_cfgs.{actor,critic}.{activation = relu, hidden_sizes = [H, H]} through the Agent facade (the reference's
    efficiency table has a 1024 x 1024 row, docs/source/start/efficiency.rst:23; its YAMLs use 64 x 64 tanh): the
    update runs on the per-step kernels (the persistent passes are 64-wide tanh only and decline), rollout / GAE /
    trust-region machinery / logging unchanged."""
    import omnisafe_amd

    cfg = {'seed': 2, 'train_cfgs': {'device': DEV, 'total_steps': 2 * 128 * 32, 'vector_env_nums': 128},
           'algo_cfgs': {'steps_per_epoch': 128 * 32, 'update_iters': 2},
           'model_cfgs': {'actor': {'activation': 'relu', 'hidden_sizes': [width, width]},
                          'critic': {'activation': 'relu', 'hidden_sizes': [width, width]}},
           'logger_cfgs': {'log_dir': str(tmp_path), 'verbose': False}, 'env_cfgs': {'horizon': 16, 'cost_p': 0.3}}
    agent = omnisafe_amd.Agent(algo_name, 'SynthCarGoal1-v0', custom_cfgs=cfg)
    ac = agent.agent._actor_critic
    assert ac.activation == 'relu' and ac.hidden == width | (1 << 16)
    p0 = ac.params.clone()
    ep_ret, ep_cost, ep_len = agent.learn()
    assert ep_len == 16.0 and 2.0 < ep_cost < 8.0 and np.isfinite(ep_ret)
    assert agent.agent._updater.last_path == 'per-step'
    assert torch.isfinite(ac.params).all() and not torch.equal(ac.params, p0)
