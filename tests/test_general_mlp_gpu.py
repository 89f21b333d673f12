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


@pytest.mark.parametrize('B', [300, 64])  # (64: the YAML batch_size -- the skinny kernels; 300: the tiled GEMM)
@pytest.mark.parametrize('obs_dim,act_dim,a_h,c_h,a_act,c_act', SHAPES)
def test_policy_step_kl_and_one_update_vs_oracle(monkeypatch, obs_dim, act_dim, a_h, c_h, a_act, c_act, B):
    from omnisafe_amd.update import PPOUpdater

    monkeypatch.setenv('OSA_FORCE_GENERAL_MLP', '1')
    torch.manual_seed(2)
    ac, ref = make(obs_dim, act_dim, a_h, c_h, a_act, c_act)
    assert ac.general
    M = 700  # ragged: not a multiple of any tile
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
            assert bad.mean() < 2e-3 and np.abs(got - want).max() < 1e-3, (net, key, bad.mean(), np.abs(got - want).max())
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


def _mb_call(up, dev, idx, B, lam, stats_row, mode):
    """osa_gmlp_minibatch with an explicit mode (0 clip + Adam, 1 clipped gradient only, 2 raw gradient)."""
    import ctypes as C

    from omnisafe_amd import _lib

    ac, lib = up.ac, up.lib
    ws, nws = ac.gmlp_ws(B)
    _lib.check(lib.osa_gmlp_minibatch(
        C.byref(ac.desc), _lib.ptr(ac.params), _lib.ptr(ac.adam_m), _lib.ptr(ac.adam_v), _lib.ptr(ac.adam_step),
        _lib.ptr(ac.grads), _lib.ptr(dev['obs']), dev['obs'].stride(0), _lib.ptr(dev['act']), dev['act'].stride(0),
        _lib.ptr(dev['logp']), _lib.ptr(dev['target_value_r']), _lib.ptr(dev['target_value_c']), _lib.ptr(dev['adv_r']),
        _lib.ptr(dev['adv_c']), _lib.ptr(idx), B, _lib.ptr(lam), C.byref(up.hp), up.loss_kind, mode, up._nets_mask(),
        None, 0.0, _lib.ptr(ws), nws, _lib.ptr(stats_row), _lib.stream_ptr()), 'osa_gmlp_minibatch')


@pytest.mark.parametrize('B', [64, 37, 3])
@pytest.mark.parametrize('obs_dim,act_dim,a_h,c_h,a_act,c_act', SHAPES + [
    (60, 2, [100, 100, 100, 100, 100, 100, 100], [20], 'sigmoid', 'softplus'),  # seven hidden layers / one
    (33, 5, [], [], 'tanh', 'tanh'),                                            # no hidden layer at all
    (61, 3, [30, 7], [50, 1, 9], 'tanh', 'relu')])                              # widths that are not multiples of 4 (row padding)
def test_skinny_step_equals_the_tiled_step(monkeypatch, obs_dim, act_dim, a_h, c_h, a_act, c_act, B):
    """Minibatches of <= 64 rows (the YAML batch_size) take the skinny kernels (general_mlp.hip: gs_fwd / gs_bwd /
    gs_wgrad -- weights streamed once per pass, clip + Adam from recomputed gradient tiles); OSA_GMLP_SKINNY=0 keeps
    them on the tiled GEMM path that the tests above pin to the oracle and to the reference's goldens.  Same inputs,
    three chained optimiser steps, then one clipped and one raw gradient: parameters, moments, statistics and
    gradients agree to the summation order of float32 MFMA chains."""
    from omnisafe_amd.update import PPOUpdater

    monkeypatch.setenv('OSA_FORCE_GENERAL_MLP', '1')
    M = 500
    g = torch.Generator().manual_seed(5)
    cpu = {'obs': torch.randn(M, obs_dim, generator=g), 'act': torch.randn(M, act_dim, generator=g),
           'logp': -1.0 + 0.3 * torch.randn(M, generator=g), 'target_value_r': torch.randn(M, generator=g),
           'target_value_c': torch.randn(M, generator=g), 'adv_r': torch.randn(M, generator=g),
           'adv_c': torch.randn(M, generator=g)}
    dev = {k: v.to(DEV).contiguous() for k, v in cpu.items()}
    idxs = [torch.randperm(M, generator=g)[:B].to(DEV) for _ in range(5)]
    lam = torch.tensor([0.7], device=DEV)
    out = {}
    for skinny in ('1', '0'):
        monkeypatch.setenv('OSA_GMLP_SKINNY', skinny)
        torch.manual_seed(3)
        ac, _ = make(obs_dim, act_dim, a_h, c_h, a_act, c_act)
        up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False, entropy_coef=0.02)
        up.hp.lr_actor, up.hp.lr_critic = 3e-4, 1e-3
        stats = torch.zeros(5, 16, device=DEV)
        for k in range(3):
            up.minibatch(dev, idxs[k], B, lam, stats[k])
        res = {'params': ac.params.clone(), 'm': ac.adam_m.clone(), 'v': ac.adam_v.clone(),
               'step': ac.adam_step.clone()}
        _mb_call(up, dev, idxs[3], B, lam, stats[3], 1)
        res['g_clipped'] = ac.grads.clone()
        _mb_call(up, dev, idxs[4], B, lam, stats[4], 2)
        res['g_raw'] = ac.grads.clone()
        assert torch.equal(ac.params, res['params']) and torch.equal(ac.adam_step, res['step'])  # modes 1 / 2: no update
        res['stats'] = stats.clone()
        out[skinny] = res
    a, b = out['1'], out['0']
    assert a['step'].tolist() == [3, 3, 3] == b['step'].tolist()
    for key in ('g_raw', 'g_clipped', 'm', 'v'):
        x, y = a[key].cpu().numpy(), b[key].cpu().numpy()
        scale = np.abs(y).max() + 1e-30
        assert np.abs(x - y).max() <= 1e-4 * scale, (key, np.abs(x - y).max(), scale)
    x, y = a['params'].cpu().numpy(), b['params'].cpu().numpy()
    # (three Adam steps of lr <= 1e-3: elements whose gradient is of the order of eps amplify summation-order noise)
    bad = np.abs(x - y) > 2e-6 + 1e-4 * np.abs(y)
    assert bad.mean() < 2e-3 and np.abs(x - y).max() < 1e-3, (bad.mean(), np.abs(x - y).max())
    sa, sb = a['stats'].cpu().numpy(), b['stats'].cpu().numpy()
    np.testing.assert_allclose(sa[:, :10], sb[:, :10], rtol=2e-4, atol=1e-6)
    # zero padding of the blocks stays zero on the skinny path as well
    real = torch.zeros_like(ac.params, dtype=torch.bool)
    for i, net in enumerate((ac.actor, ac.reward_critic, ac.cost_critic)):
        real[i, net._flat_index] = True
    for key in ('params', 'm', 'v', 'g_raw', 'g_clipped'):
        assert float(a[key][~real].abs().max()) == 0.0, key


HIDDEN = [  # tests/golden/<tag>.npz = one `_update()` of the UNMODIFIED reference with these model_cfgs
    ('hidden256x128_ppolag_point', 'PPOLag', 'SynthPointGoal1-v0', ({}, {'lagrangian_multiplier_init': 0.5, 'cost_limit': 0.5}),
     {'actor': {'hidden_sizes': [256, 128]}, 'critic': {'hidden_sizes': [256, 128]}}),
    ('hidden256x128_focops_point', 'FOCOPS', 'SynthPointGoal1-v0', ({'focops_eta': 0.02}, {'cost_limit': 1.0}),
     {'actor': {'hidden_sizes': [256, 128]}, 'critic': {'hidden_sizes': [256, 128]}}),
    ('hidden96x40x24_p3o_point', 'P3O', 'SynthPointGoal1-v0', ({'cost_limit': 0.5, 'kappa': 2.0}, None),
     {'actor': {'hidden_sizes': [96, 40, 24]}, 'critic': {'hidden_sizes': [80, 48]}}),
]


def _run_hidden(name, g, tmp_path, env_id, extra, model, batch_size):
    import omnisafe_amd

    N, T = int(g['N']), int(g['T'])
    extra_algo, lag = extra
    cfg = {'seed': 0, 'train_cfgs': {'device': DEV, 'total_steps': 4 * N * T, 'vector_env_nums': N},
           'algo_cfgs': dict({'steps_per_epoch': N * T, 'update_iters': 2, 'kl_early_stop': False,
                              'batch_size': batch_size}, **extra_algo),
           'model_cfgs': model, 'logger_cfgs': {'log_dir': str(tmp_path), 'verbose': False}}
    if lag:
        cfg['lagrange_cfgs'] = lag
    algo = omnisafe_amd.Agent(name, env_id, custom_cfgs=cfg).agent
    ac = algo._actor_critic
    assert ac.general
    for net in ('actor', 'reward_critic', 'cost_critic'):
        sd = {k[len('init/') + len(net) + 1:]: torch.from_numpy(v.copy()) for k, v in g.items()
              if k.startswith(f'init/{net}/')}
        getattr(ac, net).load_state_dict(sd)
    data = {k[5:]: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in g.items() if k.startswith('data/')}
    algo._buf.get = lambda: dict(data)
    algo._logger.extend('Metrics/EpCost', [float(v) for v in g['ep_cost_window']])
    algo._perms_override = [torch.from_numpy(p.copy()) for p in g['perms']]
    algo._update()
    return algo, ac


def _hidden_check(ac, g, nets, atol, stragglers=0):
    """|theta - theta_reference| <= atol; `stragglers` elements per tensor may reach 1.5e-5 (Adam's first steps are
    lr g / (|g| + eps): where a gradient element is of the order of eps the summation order of a float32 contraction
    decides its sign -- 1 of 32 768 elements of one tensor in the recordings, as in tests/test_dp_golden_gpu.py)."""
    for net in nets:
        for k, v in getattr(ac, net).state_dict().items():
            err = np.abs(v.cpu().numpy() - g[f'post/{net}/{k}'])
            assert int((err > atol).sum()) <= stragglers and float(err.max()) <= (1.5e-5 if stragglers else atol), (
                net, k, int((err > atol).sum()), float(err.max()))


@pytest.mark.parametrize('skinny', ['1', '0'])
@pytest.mark.parametrize('tag,name,env_id,extra,model', HIDDEN)
def test_first_order_update_of_the_reference_at_general_widths(golden, tmp_path, monkeypatch, tag, name, env_id, extra,
                                                              model, skinny):
    """One whole `_update()` of the unmodified reference built with `hidden_sizes` OUTSIDE the fused family
    (oracle/make_golden.py::gen_hidden_shape_updates: [256, 128] for PPOLag and FOCOPS, actor [96, 40, 24] / critics
    [80, 48] for P3O with its penalty active): 32 chained 64-row Adam steps per network on the layer-wise path -- skinny
    kernels and tiled GEMM -- against the reference's post-update parameters, at the tolerance of the fused kernels'
    config-shape tests (atol 2e-6; the parameters move by > 5e-3)."""
    monkeypatch.setenv('OSA_GMLP_SKINNY', skinny)
    g = golden(f'{tag}.npz')
    algo, ac = _run_hidden(name, g, tmp_path, env_id, extra, model, 64)
    assert algo._last_update_steps == 2 * 16
    err = {n: max(float(np.abs(v.cpu().numpy() - g[f'post/{n}/{k}']).max()) for k, v in getattr(ac, n).state_dict().items())
           for n in ('actor', 'reward_critic', 'cost_critic')}
    print(tag, 'skinny', skinny, 'max |param - reference|:', err)
    _hidden_check(ac, g, ('actor', 'reward_critic', 'cost_critic'), 2e-6, stragglers=2)
    moved = max(float(np.abs(g[f'post/actor/{k}'] - g[f'init/actor/{k}']).max()) for k in ac.actor.state_dict())
    assert moved > 5e-3
    log = lambda key: np.asarray(list(algo._logger._data[key]), np.float64)  # noqa: E731
    np.testing.assert_allclose(log('Train/KL')[-1], g['log/Train/KL'][-1], rtol=1e-2, atol=1e-7)
    np.testing.assert_allclose(log('Loss/Loss_pi').mean(), g['log/Loss/Loss_pi'].mean(), rtol=2e-3, atol=2e-6)
    for key in ('Loss/Loss_reward_critic', 'Loss/Loss_cost_critic'):
        np.testing.assert_allclose(log(key).mean(), g['log/' + key].mean(), rtol=2e-4)
    if name == 'P3O':
        assert float(g['log/Loss/Loss_pi_cost'].mean()) > 0.1  # the penalty is active in this recording
        np.testing.assert_allclose(log('Loss/Loss_pi_cost')[-1], g['log/Loss/Loss_pi_cost'].mean(), rtol=1e-3)
    if 'lambda_after' in g:
        np.testing.assert_allclose(algo._lagrange.lagrangian_multiplier, float(g['lambda_after']), rtol=1e-6)


def test_trpolag_update_of_the_reference_at_general_widths(golden, tmp_path):
    """TRPOLag with hidden_sizes [256, 128] on 27 / 8 (tests/golden/hidden256x128_trpolag_ant.npz): Fisher-vector
    products, CG, line search and the 128-row critic passes of the unmodified reference on the layer-wise path --
    accepted line-search index identical, tolerances of tests/test_config_shapes_gpu.py's trust-region family."""
    g = golden('hidden256x128_trpolag_ant.npz')
    model = {'actor': {'hidden_sizes': [256, 128]}, 'critic': {'hidden_sizes': [256, 128]}}
    algo, ac = _run_hidden('TRPOLag', g, tmp_path, 'SynthAnt-v0', ({}, {'lagrangian_multiplier_init': 0.5, 'cost_limit': 0.5}),
                           model, 128)
    log = lambda key: np.asarray(list(algo._logger._data[key]), np.float64)  # noqa: E731
    assert int(log('Misc/AcceptanceStep')[-1]) == int(g['log/Misc/AcceptanceStep'][-1])
    for key, rtol in (('Misc/Alpha', 1e-2), ('Misc/xHx', 1e-2), ('Misc/gradient_norm', 1e-3), ('Misc/H_inv_g', 1e-2),
                      ('Misc/FinalStepNorm', 2e-2)):
        np.testing.assert_allclose(log(key)[-1], g['log/' + key][-1], rtol=rtol, err_msg=key)
    np.testing.assert_allclose(algo._lagrange.lagrangian_multiplier, float(g['lambda_after']), rtol=1e-6)
    _hidden_check(ac, g, ('actor',), 5e-5)
    _hidden_check(ac, g, ('reward_critic', 'cost_critic'), 2e-5)


@pytest.fixture(scope='module')
def live_1024(tmp_path_factory):
    """One `_update()` of the UNMODIFIED reference with hidden_sizes [1024, 1024] on all three networks -- the shape of
    the reference's one published timing table (docs/source/start/efficiency.rst:15-23; builder utils/model.py:73-111) --
    recorded NOW on this box's host by the committed generator in its own process (`oracle/make_golden.py hidden-shapes
    hidden1024x1024_ppolag_point`; from /root/reference in the build container, from the staged archive
    oracle/_ref/omnisafe_ref.zip on the GPU box).  27 MB: a live recording instead of a committed fixture."""
    import os
    import subprocess
    import sys

    import ref_harness

    if not ref_harness.reference_available():
        pytest.skip('no reference: neither /root/reference nor oracle/_ref/omnisafe_ref.zip')
    out = tmp_path_factory.mktemp('live1024')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OSA_GOLDEN_OUT=str(out), OMP_NUM_THREADS='8')
    env.pop('OSA_FORCE_GENERAL_MLP', None)
    p = subprocess.run([sys.executable, os.path.join(root, 'oracle', 'make_golden.py'), 'hidden-shapes',
                        'hidden1024x1024_ppolag_point'], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    g = dict(np.load(out / 'hidden1024x1024_ppolag_point.npz'))
    assert g['init/actor/mean.2.weight'].shape == (1024, 1024) and g['init/reward_critic/critic_0.2.weight'].shape == (1024, 1024)
    return g


@pytest.mark.reference
@pytest.mark.parametrize('skinny', ['1', '0'])
def test_update_of_the_live_reference_at_1024x1024(live_1024, tmp_path, monkeypatch, skinny):
    """The skinny kernels (OSA_GMLP_SKINNY=1) and the tiled GEMM path (= 0) against the reference itself at
    1024 x 1024: 32 chained 64-row Adam steps of three 1.1 M-parameter networks.  Tolerance as the committed
    hidden-shape recordings (atol 2e-6), with stragglers in proportion: a 1024-term float32 contraction in MFMA-tile
    order v the CPU's sgemm order moves a gradient element by ~1e-7 relative, and Adam's first steps turn an element
    whose gradient is of the order of eps into a visible fraction of lr -- 2 of 32 768 elements per tensor in the
    [256, 128] recordings, allowed here: 64 of 1 048 576, none beyond 1.5e-5 (a twentieth of one learning-rate step)."""
    monkeypatch.setenv('OSA_GMLP_SKINNY', skinny)
    g = live_1024
    model = {'actor': {'hidden_sizes': [1024, 1024]}, 'critic': {'hidden_sizes': [1024, 1024]}}
    algo, ac = _run_hidden('PPOLag', g, tmp_path, 'SynthPointGoal1-v0',
                           ({}, {'lagrangian_multiplier_init': 0.5, 'cost_limit': 0.5}), model, 64)
    assert algo._last_update_steps == 2 * 16
    worst = {}
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in getattr(ac, net).state_dict().items():
            err = np.abs(v.cpu().numpy() - g[f'post/{net}/{k}'])
            worst[f'{net}/{k}'] = (int((err > 2e-6).sum()), float(err.max()), err.size)
    print('skinny', skinny, '(elements > 2e-6, max |param - live reference|, size):', worst)
    for key, (n_big, mx, size) in worst.items():
        assert n_big <= max(2, size // 16384) and mx <= 1.5e-5, (key, n_big, mx, size)
    moved = max(float(np.abs(g[f'post/actor/{k}'] - g[f'init/actor/{k}']).max()) for k in ac.actor.state_dict())
    assert moved > 5e-3
    log = lambda key: np.asarray(list(algo._logger._data[key]), np.float64)  # noqa: E731
    np.testing.assert_allclose(log('Train/KL')[-1], g['log/Train/KL'][-1], rtol=1e-2, atol=1e-7)
    for key in ('Loss/Loss_reward_critic', 'Loss/Loss_cost_critic'):
        np.testing.assert_allclose(log(key).mean(), g['log/' + key].mean(), rtol=2e-4)
    np.testing.assert_allclose(algo._lagrange.lagrangian_multiplier, float(g['lambda_after']), rtol=1e-6)


@pytest.mark.parametrize('mask_kind', ['critics-only', 'actor-only', 'no-cost'])
def test_skinny_step_with_network_masks_equals_the_tiled_step(monkeypatch, mask_kind):
    """The trust-region family updates the critics alone in its minibatch loop (natural_pg.py:205-223), PPO / PolicyGradient
    leave the cost critic out (use_cost False): the skinny kernels under the same network masks as the tiled path --
    untouched networks stay bit-for-bit untouched, the updated ones agree as in test_skinny_step_equals_the_tiled_step."""
    from omnisafe_amd.update import PPOUpdater

    monkeypatch.setenv('OSA_FORCE_GENERAL_MLP', '1')
    M, B, obs_dim, act_dim = 300, 64, 27, 8
    g = torch.Generator().manual_seed(8)
    cpu = {'obs': torch.randn(M, obs_dim, generator=g), 'act': torch.randn(M, act_dim, generator=g),
           'logp': -3.0 + 0.3 * torch.randn(M, generator=g), 'target_value_r': torch.randn(M, generator=g),
           'target_value_c': torch.randn(M, generator=g), 'adv_r': torch.randn(M, generator=g),
           'adv_c': torch.randn(M, generator=g)}
    dev = {k: v.to(DEV).contiguous() for k, v in cpu.items()}
    idxs = [torch.randperm(M, generator=g)[:B].to(DEV) for _ in range(2)]
    lam = torch.tensor([0.3], device=DEV)
    kw = {'critics-only': {'update_actor': False}, 'actor-only': {'update_critics': False}, 'no-cost': {'use_cost': False}}[mask_kind]
    out = {}
    for skinny in ('1', '0'):
        monkeypatch.setenv('OSA_GMLP_SKINNY', skinny)
        torch.manual_seed(4)
        ac, _ = make(obs_dim, act_dim, [96, 48], [80], 'tanh', 'relu')
        p0 = ac.params.clone()
        up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False, **kw)
        up.hp.lr_actor, up.hp.lr_critic = 3e-4, 1e-3
        stats = torch.zeros(2, 16, device=DEV)
        for k in range(2):
            up.minibatch(dev, idxs[k], B, lam, stats[k])
        out[skinny] = (ac.params.clone(), ac.adam_step.clone(), p0)
    (pa, sa, p0), (pb, sb, _) = out['1'], out['0']
    assert sa.tolist() == sb.tolist()
    untouched = {'critics-only': [0], 'actor-only': [1, 2], 'no-cost': [2]}[mask_kind]
    for net in range(3):
        if net in untouched:
            assert torch.equal(pa[net], p0[net]) and sa[net].item() == 0, net
        else:
            assert sa[net].item() == 2 and not torch.equal(pa[net], p0[net])
            x, y = pa[net].cpu().numpy(), pb[net].cpu().numpy()
            bad = np.abs(x - y) > 2e-6 + 1e-4 * np.abs(y)
            assert bad.mean() < 2e-3 and np.abs(x - y).max() < 1e-3, (net, bad.mean(), np.abs(x - y).max())


@pytest.mark.parametrize('B', [64, 41])
@pytest.mark.parametrize('obs_dim,act_dim,a_h,c_h,a_act,c_act,max_norm', [
    (60, 2, [256, 128], [256, 128], 'tanh', 'tanh', 0.05),       # clip active in every network, L2 term on
    (376, 17, [160, 96], [96], 'tanh', 'relu', 0.3),             # wide observations; a 17-wide top layer (own launch)
    (27, 8, [64, 80, 48, 33], [50, 1, 9], 'sigmoid', 'tanh', 40.0),  # deeper actor than critics, widths off multiples of 4
    (33, 5, [1100], [70], 'tanh', 'tanh', 0.2)])                 # more than 64 column tiles of partial outputs
def test_fused_clip_norm_equals_the_weight_gradient_launch(monkeypatch, obs_dim, act_dim, a_h, c_h, a_act, c_act,
                                                           max_norm, B):
    """Round 6: in mode 0 the clip norm of a skinny step comes out of the forward / top / backward launches
    (skinny_mlp.h: gs_gram_norm -- |dZ^T H|^2 = <dZ dZ^T, H H^T> per 16 rows of every layer, the critics' L2 cross term
    from the pre-bias outputs, |W|^2 from the forward launches) and the top layer's forward pass rides in the launch below
    it; OSA_GMLP_NORM_FUSE=0 keeps round 5's launches, whose gs_wgrad_kernel<0> forms every gradient tile for the norm.
    Same inputs, four chained steps with the clip ACTIVE (small max_grad_norm) and the critics' L2 term on: the logged
    norms (stats 7..9), the critics' |W|^2 (stats 5, 6), losses, parameters and moments agree to float32 summation
    order (reference: torch.nn.utils.clip_grad_norm_ over the same gradients, policy_gradient.py:437-442)."""
    from omnisafe_amd.update import PPOUpdater

    monkeypatch.setenv('OSA_FORCE_GENERAL_MLP', '1')
    M = 400
    g = torch.Generator().manual_seed(11)
    cpu = {'obs': torch.randn(M, obs_dim, generator=g), 'act': torch.randn(M, act_dim, generator=g),
           'logp': -1.0 + 0.3 * torch.randn(M, generator=g), 'target_value_r': torch.randn(M, generator=g),
           'target_value_c': torch.randn(M, generator=g), 'adv_r': torch.randn(M, generator=g),
           'adv_c': torch.randn(M, generator=g)}
    dev = {k: v.to(DEV).contiguous() for k, v in cpu.items()}
    idxs = [torch.randperm(M, generator=g)[:B].to(DEV) for _ in range(4)]
    lam = torch.tensor([0.4], device=DEV)
    out = {}
    for fuse in ('1', '0'):
        monkeypatch.setenv('OSA_GMLP_NORM_FUSE', fuse)
        torch.manual_seed(6)
        ac, _ = make(obs_dim, act_dim, a_h, c_h, a_act, c_act)
        up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False, entropy_coef=0.01)
        up.hp.lr_actor, up.hp.lr_critic = 3e-4, 1e-3
        up.hp.max_grad_norm = max_norm
        up.hp.use_critic_norm, up.hp.critic_norm_coef = 1, 0.01
        stats = torch.zeros(4, 16, device=DEV)
        for k in range(4):
            up.minibatch(dev, idxs[k], B, lam, stats[k])
        out[fuse] = {'params': ac.params.clone(), 'm': ac.adam_m.clone(), 'v': ac.adam_v.clone(),
                     'step': ac.adam_step.clone(), 'stats': stats.clone()}
    a, b = out['1'], out['0']
    assert a['step'].tolist() == [4, 4, 4] == b['step'].tolist()
    sa, sb = a['stats'].cpu().numpy(), b['stats'].cpu().numpy()
    assert (sb[:, 8:10] > max_norm).all() or max_norm >= 40.0    # the clip is active (critics) where the case says so
    np.testing.assert_allclose(sa[:, 7:10], sb[:, 7:10], rtol=2e-5)    # total gradient norms
    np.testing.assert_allclose(sa[:, 5:7], sb[:, 5:7], rtol=2e-5)      # the critics' sum of squared parameters
    np.testing.assert_allclose(sa[:, :5], sb[:, :5], rtol=2e-4, atol=1e-6)
    for key in ('m', 'v'):
        x, y = a[key].cpu().numpy(), b[key].cpu().numpy()
        scale = np.abs(y).max() + 1e-30
        assert np.abs(x - y).max() <= 1e-4 * scale, (key, np.abs(x - y).max(), scale)
    x, y = a['params'].cpu().numpy(), b['params'].cpu().numpy()
    bad = np.abs(x - y) > 2e-6 + 1e-4 * np.abs(y)
    assert bad.mean() < 2e-3 and np.abs(x - y).max() < 1e-3, (bad.mean(), np.abs(x - y).max())
