"""GPU: the layer-wise path for GENERAL network shapes (csrc/general_mlp.hip; any `hidden_sizes` the reference's
model builder accepts, omnisafe/utils/model.py:73-111 -- the shapes models.py used to refuse), against

  * the oracle's torch modules (oracle/np_oracle.py: Actor / Critic built by the reference's recipe, critic_step /
    actor_step / kl_old_new / fvp restating policy_gradient.py:428-524, natural_pg.py:91-119) at shapes from one
    hidden layer to 1024 x 1024 (docs/source/start/efficiency.rst:15-23), actor != critic widths, relu / tanh;
  * the unmodified reference's goldens: OSA_FORCE_GENERAL_MLP=1 sends the YAML-default [64, 64] networks through the
    general path, so one whole `_update()` of PPOLag (60 / 2), CPO (72 / 2) and TRPOLag (27 / 8) is pinned to the
    reference exactly as for the fused kernels (tests/golden/config*_*.npz).

Tolerances: float32 MFMA chains against the CPU's sgemm order: parameters after one Adam step rtol 1e-4 / atol 2e-6
(as test_mlp_gpu.py::test_large_batch_multiblock_equals_single_pass), Fisher-vector product rel. 2e-3, goldens as
tests/test_config_shapes_gpu.py."""
import types

import numpy as np
import pytest
import torch

import np_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def cfgs(a_h, c_h, a_act='tanh', c_act='tanh', lr=3e-4):
    ns = types.SimpleNamespace
    return ns(actor=ns(hidden_sizes=list(a_h), activation=a_act, lr=lr),
              critic=ns(hidden_sizes=list(c_h), activation=c_act, lr=lr),
              weight_initialization_mode='kaiming_uniform', actor_type='gaussian_learning', linear_lr_decay=True)


def make(obs_dim, act_dim, a_h, c_h, a_act='tanh', c_act='tanh'):
    """(general device actor-critic, oracle modules with the same parameters)"""
    from omnisafe_amd.models import ConstraintActorCritic
    from omnisafe_amd.spaces import Box

    ac = ConstraintActorCritic(Box(-np.inf, np.inf, (obs_dim,)), Box(-1, 1, (act_dim,)), cfgs(a_h, c_h, a_act, c_act),
                               4, device=DEV)
    ref = types.SimpleNamespace(actor=O.Actor(obs_dim, act_dim, tuple(a_h), a_act),
                                reward_critic=O.Critic(obs_dim, tuple(c_h), c_act),
                                cost_critic=O.Critic(obs_dim, tuple(c_h), c_act))
    with torch.no_grad():
        ref.actor.log_std.copy_(torch.tensor(np.linspace(-0.4, 0.3, act_dim), dtype=torch.float32))
    for net in ('actor', 'reward_critic', 'cost_critic'):
        getattr(ac, net).load_state_dict(getattr(ref, net).state_dict())
    return ac, ref


SHAPES = [  # obs, act, actor hidden, critic hidden, actor act, critic act
    (60, 2, [64, 64], [64, 64], 'tanh', 'tanh'),        # fused-family shape (forced through the general path below)
    (60, 2, [256, 128], [256, 128], 'tanh', 'tanh'),    # unequal widths
    (27, 8, [64], [64], 'relu', 'relu'),                # ONE hidden layer, unaligned observation rows
    (72, 2, [32, 48, 16], [40, 24], 'tanh', 'relu'),    # three / two hidden layers, actor != critics
    (60, 2, [1024, 1024], [1024, 1024], 'tanh', 'tanh'),  # the reference's published large-network row
    (376, 17, [300, 200], [300, 200], 'tanh', 'tanh'),  # Humanoid-sized input, widths that are not tile multiples
]


def test_models_accept_what_the_reference_builds(monkeypatch):
    """`general` is chosen exactly for the shapes outside the fused family; state_dict keys / shapes are the reference's."""
    ac, ref = make(60, 2, [64, 64], [64, 64])
    assert not ac.general
    for a_h, c_h in (([256, 128], [256, 128]), ([64], [64]), ([64, 64, 64], [64, 64]), ([1024, 1024], [1024, 1024]),
                     ([48, 48], [48, 48]), ([], [])):
        ac, ref = make(60, 2, a_h, c_h)
        assert ac.general and ac.hidden == 0
        assert list(ac.actor.state_dict()) == list(ref.actor.state_dict())
        for net in ('actor', 'reward_critic', 'cost_critic'):
            for (k, v), (k2, v2) in zip(getattr(ac, net).state_dict().items(), getattr(ref, net).state_dict().items()):
                assert k == k2 and tuple(v.shape) == tuple(v2.shape) and torch.equal(v.cpu(), v2)
    monkeypatch.setenv('OSA_FORCE_GENERAL_MLP', '1')
    ac, _ = make(60, 2, [64, 64], [64, 64])
    assert ac.general


@pytest.mark.parametrize('obs_dim,act_dim,a_h,c_h,a_act,c_act', SHAPES)
def test_policy_step_kl_and_one_update_vs_oracle(monkeypatch, obs_dim, act_dim, a_h, c_h, a_act, c_act):
    from omnisafe_amd.update import PPOUpdater

    monkeypatch.setenv('OSA_FORCE_GENERAL_MLP', '1')
    torch.manual_seed(2)
    ac, ref = make(obs_dim, act_dim, a_h, c_h, a_act, c_act)
    assert ac.general
    M, B = 700, 300  # ragged: not a multiple of any tile
    obs = torch.randn(M, obs_dim)
    eps = torch.randn(M, act_dim)
    # ---- ConstraintActorCritic.step (constraint_actor_critic.py:84-109)
    with torch.no_grad():
        d = ref.actor.dist(obs)
        act_ref = d.mean + eps * d.stddev
        logp_ref = d.log_prob(act_ref).sum(-1)
        vr_ref, vc_ref = ref.reward_critic(obs), ref.cost_critic(obs)
    act, v_r, v_c, logp = ac.step(obs.to(DEV), eps=eps.to(DEV))
    np.testing.assert_allclose(act.cpu().numpy(), act_ref.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(v_r.cpu().numpy(), vr_ref.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(v_c.cpu().numpy(), vc_ref.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(logp.cpu().numpy(), logp_ref.numpy(), rtol=1e-4, atol=1e-4)
    a_det, _, _, _ = ac.step(obs.to(DEV), deterministic=True)
    np.testing.assert_allclose(a_det.cpu().numpy(), d.mean.numpy(), rtol=1e-4, atol=2e-5)
    # ---- one optimiser step on a gathered minibatch (policy_gradient.py:428-445, 468-485, 514-524)
    cpu = {'obs': obs, 'act': act_ref, 'logp': logp_ref + 0.3 * torch.randn(M), 'target_value_r': torch.randn(M),
           'target_value_c': torch.randn(M), 'adv_r': torch.randn(M), 'adv_c': torch.randn(M)}
    dev = {k: v.to(DEV).contiguous() for k, v in cpu.items()}
    idx = torch.randperm(M)[:B]
    lam = 0.5
    opt = {n: torch.optim.Adam(getattr(ref, n).parameters(), lr=lr) for n, lr in
           (('actor', 3e-4), ('reward_critic', 1e-3), ('cost_critic', 1e-3))}
    up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False, entropy_coef=0.02)
    up.hp.lr_actor, up.hp.lr_critic = 3e-4, 1e-3
    up.snapshot_old_distribution(dev['obs'])
    with torch.no_grad():
        old = ref.actor.dist(obs)
        om, osd = old.mean.clone(), old.stddev.clone()
    stats = torch.zeros(2, 16, device=DEV)
    for k in range(2):  # two steps: the second one sees Adam's moments
        sub = {key: v[idx] for key, v in cpu.items()}
        l_r = O.critic_step(ref.reward_critic, opt['reward_critic'], sub['obs'], sub['target_value_r'])
        l_c = O.critic_step(ref.cost_critic, opt['cost_critic'], sub['obs'], sub['target_value_c'])
        l_p, ent, ratio = O.actor_step(ref.actor, opt['actor'], sub['obs'], sub['act'], sub['logp'], sub['adv_r'],
                                       sub['adv_c'], lam, entropy_coef=0.02)
        up.minibatch(dev, idx.to(DEV), B, torch.tensor([lam], device=DEV), stats[k])
        s = stats[k].cpu().numpy()
        np.testing.assert_allclose(s[0] + 0.001 * s[5], l_r, rtol=2e-4)
        np.testing.assert_allclose(s[1] + 0.001 * s[6], l_c, rtol=2e-4)
        np.testing.assert_allclose(s[2], l_p, rtol=2e-3, atol=2e-5)
        np.testing.assert_allclose(s[3], float(ratio.mean()), rtol=2e-4)
    assert ac.adam_step.tolist() == [2, 2, 2]
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for key, v in getattr(ac, net).state_dict().items():
            want = getattr(ref, net).state_dict()[key].numpy()
            got = v.cpu().numpy()
            # (Adam's first steps are lr g / (|g| + eps): elements whose gradient is of the order of eps amplify
            # summation-order differences -- all but a handful agree to float32 rounding)
            bad = np.abs(got - want) > 2e-6 + 1e-4 * np.abs(want)
            assert bad.mean() < 2e-3 and np.abs(got - want).max() < 3e-3, (net, key, bad.mean(), np.abs(got - want).max())
    # ---- KL(old || new) after the two steps (policy_gradient.py:383-390)
    kl = float(up.kl(dev['obs']))
    np.testing.assert_allclose(kl, O.kl_old_new(ref.actor, obs, om, osd), rtol=5e-3, atol=1e-7)
    # padding of the parameter blocks stayed zero (it is part of the clip norm)
    real = torch.zeros_like(ac.params, dtype=torch.bool)
    for i, net in enumerate((ac.actor, ac.reward_critic, ac.cost_critic)):
        real[i, net._flat_index] = True
    assert float(ac.params[~real].abs().max()) == 0.0 and float(ac.adam_m[~real].abs().max()) == 0.0


@pytest.mark.parametrize('obs_dim,act_dim,a_h,a_act', [(72, 2, [64, 64], 'tanh'), (60, 2, [256, 128], 'tanh'),
                                                        (27, 8, [64], 'relu'), (60, 3, [96, 40, 24], 'tanh')])
def test_fvp_gradient_and_line_search_evaluation_vs_oracle(monkeypatch, obs_dim, act_dim, a_h, a_act):
    """NaturalPG._fvp (natural_pg.py:91-119) as JVP -> VJP on the layer-wise path against the oracle's double-backward
    autograd; full-batch policy gradient (trpo.py:176-186); candidate evaluation (trpo.py:102-138)."""
    from omnisafe_amd.trust_region import TrustRegionSolver

    monkeypatch.setenv('OSA_FORCE_GENERAL_MLP', '1')
    torch.manual_seed(7)
    ac, ref = make(obs_dim, act_dim, a_h, a_h, a_act, a_act)
    M = 1500
    obs = torch.randn(M, obs_dim)
    s = TrustRegionSolver(ac, cg_iters=10, cg_damping=0.1)
    s.begin(obs.to(DEV))
    v = torch.randn(ac.actor.num_params)
    Fv = ac.actor.unpad(s.fvp(ac.actor.pad(v))).cpu().numpy()
    Fv_ref = O.fvp(ref.actor, obs, v, cg_damping=0.1).numpy()
    assert np.linalg.norm(Fv - Fv_ref) / np.linalg.norm(Fv_ref) < 2e-3
    # policy gradient of -mean(ratio * adv) and its loss
    with torch.no_grad():
        d = ref.actor.dist(obs)
        act = d.mean + torch.randn(M, act_dim) * d.stddev
        logp = d.log_prob(act).sum(-1) + 0.2 * torch.randn(M)
    adv_r, adv_c = torch.randn(M), torch.randn(M)
    data = {'obs': obs.to(DEV), 'act': act.to(DEV), 'logp': logp.to(DEV), 'adv_r': adv_r.to(DEV), 'adv_c': adv_c.to(DEV),
            'target_value_r': torch.zeros(M, device=DEV), 'target_value_c': torch.zeros(M, device=DEV)}
    lam = torch.tensor([0.4], device=DEV)
    loss, grad = s.actor_loss_grad(data, 'adv_r', 'adv_c', lam)
    ref.actor.zero_grad()
    l_ref, _, _ = O.pg_loss_pi(ref.actor, obs, act, logp, O.lag_adv_surrogate(adv_r, adv_c, 0.4))
    l_ref.backward()
    g_ref = O.flat_grads(ref.actor).numpy()
    np.testing.assert_allclose(float(loss), float(l_ref), rtol=1e-4, atol=1e-6)
    g = ac.actor.unpad(grad).cpu().numpy()
    assert np.linalg.norm(g - g_ref) / np.linalg.norm(g_ref) < 1e-4
    # candidates theta_old + frac * step: [loss_pi, loss_cost, kl, mean ratio]
    theta_old = ac.params[0].clone()
    step = ac.actor.pad(0.05 * torch.randn(ac.actor.num_params))
    res = s.evaluate_candidates(data, theta_old, step, [1.0, 0.5], 'adv_r', lam).numpy()
    with torch.no_grad():
        old = ref.actor.dist(obs)
        om, osd = old.mean.clone(), old.stddev.clone()
        th = O.flat_params(ref.actor).clone()
        for k, frac in enumerate((1.0, 0.5)):
            O.set_flat_params(ref.actor, th + frac * ac.actor.unpad(step).cpu())
            q = ref.actor.dist(obs)
            ratio = torch.exp(q.log_prob(act).sum(-1) - logp)
            want = [-(ratio * O.lag_adv_surrogate(adv_r, adv_c, 0.4)).mean(), (ratio * adv_c).mean(),
                    torch.distributions.kl.kl_divergence(torch.distributions.Normal(om, osd), q).mean(), ratio.mean()]
            np.testing.assert_allclose(res[k], [float(x) for x in want], rtol=2e-3, atol=2e-6)
    assert torch.equal(ac.params[0], theta_old)


GOLDEN_CASES = [('config2_ppolag_point', 'PPOLag', 'SynthPointGoal1-v0', False),
                ('config3_cpo_car', 'CPO', 'SynthCarGoal1-v0', True),
                ('config5_trpolag_ant', 'TRPOLag', 'SynthAnt-v0', True)]


@pytest.mark.parametrize('tag,name,env_id,trust_region', GOLDEN_CASES)
def test_general_path_reproduces_the_reference_goldens(golden, tmp_path, monkeypatch, tag, name, env_id, trust_region):
    """One whole `_update()` of the UNMODIFIED reference per algorithm family with the [64, 64] networks forced
    through the general path: the same comparison as tests/test_config_shapes_gpu.py."""
    from test_config_shapes_gpu import FIRST_ORDER, TRUST_REGION
    from test_siblings_gpu import _check_params, _log, _run_update

    monkeypatch.setenv('OSA_FORCE_GENERAL_MLP', '1')
    extra = {t: e for t, _, _, e in FIRST_ORDER + TRUST_REGION}[tag]
    g = golden(f'{tag}.npz')
    algo, ac = _run_update(name, tag, g, tmp_path, trust_region=trust_region, env_id=env_id, extra=extra)
    assert ac.general and algo._updater.last_path.startswith('general-')
    if trust_region:
        assert int(_log(algo, 'Misc/AcceptanceStep')[-1]) == int(g['log/Misc/AcceptanceStep'][-1])
        for key, rtol in (('Misc/Alpha', 1e-2), ('Misc/xHx', 1e-2), ('Misc/gradient_norm', 1e-3)):
            np.testing.assert_allclose(_log(algo, key)[-1], g['log/' + key][-1], rtol=rtol, err_msg=key)
        _check_params(ac, g, ('actor',), 5e-5)
        _check_params(ac, g, ('reward_critic', 'cost_critic'), 2e-5)
    else:
        assert algo._last_update_steps == 2 * 64
        _check_params(ac, g, ('actor', 'reward_critic', 'cost_critic'), 2e-6)
        np.testing.assert_allclose(algo._lagrange.lagrangian_multiplier, float(g['lambda_after']), rtol=1e-6)
        np.testing.assert_allclose(_log(algo, 'Train/KL')[-1], g['log/Train/KL'][-1], rtol=1e-2, atol=1e-7)
        np.testing.assert_allclose(_log(algo, 'Loss/Loss_pi').mean(), g['log/Loss/Loss_pi'].mean(), rtol=2e-3, atol=2e-6)


@pytest.mark.parametrize('algo_name,a_h,c_h', [('PPOLag', [1024, 1024], [1024, 1024]), ('TRPOLag', [256, 128], [64]),
                                               ('CPO', [64], [64, 64, 64])])
def test_agents_with_general_networks_end_to_end(tmp_path, algo_name, a_h, c_h):
    """`Agent(...).learn()` with hidden_sizes the fused family does not cover -- incl. the 1024 x 1024 networks of the
    reference's published timing table -- through the registry, the rollout, the buffer and the whole update."""
    import omnisafe_amd

    cfg = {'seed': 3, 'train_cfgs': {'device': DEV, 'total_steps': 2 * 64 * 32, 'vector_env_nums': 64},
           'algo_cfgs': {'steps_per_epoch': 64 * 32, 'update_iters': 2},
           'model_cfgs': {'actor': {'hidden_sizes': a_h}, 'critic': {'hidden_sizes': c_h}},
           'logger_cfgs': {'log_dir': str(tmp_path), 'verbose': False}, 'env_cfgs': {'horizon': 16, 'cost_p': 0.3}}
    agent = omnisafe_amd.Agent(algo_name, 'SynthCarGoal1-v0', custom_cfgs=cfg)
    a = agent.agent
    p0 = a._actor_critic.params.clone()
    ep_ret, ep_cost, ep_len = agent.learn()
    assert a._actor_critic.general and a._updater.last_path.startswith('general-')
    assert ep_len == 16.0 and 2.0 < ep_cost < 8.0 and np.isfinite(ep_ret)
    p = a._actor_critic.params
    assert bool(torch.isfinite(p).all()) and not torch.equal(p, p0)
    sd = a._actor_critic.actor.state_dict()
    assert tuple(sd['mean.0.weight'].shape) == (a_h[0], 72)
