"""GPU parity of the matrix-core actor-critic kernels against the reference's golden vectors and the
oracle.  Floating-point tolerances (float32 MFMA fma-chains vs the reference's CPU sgemm):
  forward (act / value / logp):           rtol 1e-4, atol 1e-5
  one optimiser step (loss, params):      rtol 1e-4, atol 1e-6   
  parameters after 9 chained Adam steps:  atol 5e-6 on values of magnitude <= 1 (Adam's
  m/sqrt(v) normalisation turns 1e-7 gradient noise into ~1e-3*lr parameter noise per step)."""
import types

import numpy as np
import pytest
import torch

import np_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def model_cfgs(lr=3e-4, activation='tanh', width=64):
    ns = types.SimpleNamespace
    return ns(actor=ns(hidden_sizes=[width, width], activation=activation, lr=lr),
              critic=ns(hidden_sizes=[width, width], activation=activation, lr=lr),
              weight_initialization_mode='kaiming_uniform', actor_type='gaussian_learning',
              linear_lr_decay=True)


def make_ac(obs_dim, act_dim, g=None, prefix='', epochs=4, activation='tanh', width=64):
    from omnisafe_amd.models import ConstraintActorCritic
    from omnisafe_amd.spaces import Box

    ac = ConstraintActorCritic(Box(-np.inf, np.inf, (obs_dim,)), Box(-1, 1, (act_dim,)),
                               model_cfgs(activation=activation, width=width), epochs, device=DEV)
    if g is not None:
        for net in ('actor', 'reward_critic', 'cost_critic'):
            sd = {k[len(prefix) + len(net) + 1:]: torch.from_numpy(v.copy()) for k, v in g.items()
                  if k.startswith(f'{prefix}{net}/')}
            getattr(ac, net).load_state_dict(sd)
    return ac


def test_state_dict_roundtrip_and_order(golden):
    g = golden('actor_critic_step.npz')
    ac = make_ac(60, 2, g)
    sd = ac.actor.state_dict()
    assert list(sd) == ['log_std', 'mean.0.weight', 'mean.0.bias', 'mean.2.weight', 'mean.2.bias',
                        'mean.4.weight', 'mean.4.bias']
    for k, v in sd.items():
        assert np.array_equal(v.cpu().numpy(), g[f'actor/{k}'])
    assert list(ac.reward_critic.state_dict())[0] == 'critic_0.0.weight'
    assert ac.actor.num_params == 8196 and ac.reward_critic.num_params == 8129
    flat = ac.actor.flat_params()
    ref = np.concatenate([g[f'actor/{k}'].reshape(-1) for k in sd])
    assert np.array_equal(flat.cpu().numpy(), ref)
    # padding stays zero
    total = float(ac.params.abs().sum())
    only = sum(float(getattr(ac, n).flat_params().abs().sum()) for n in ('actor', 'reward_critic', 'cost_critic'))
    assert abs(total - only) < 1e-3 * total


def test_same_seed_same_init_as_oracle_construction():
    """Identical torch seed -> identical initial weights as the reference's module construction order
    (actor, reward critic, cost critic)."""
    torch.manual_seed(123)
    ac = make_ac(60, 2)
    torch.manual_seed(123)
    ref = O.ActorCritic(60, 2)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        a, b = getattr(ac, net).state_dict(), getattr(ref, net).state_dict()
        assert list(a) == list(b)
        for k in a:
            assert np.array_equal(a[k].cpu().numpy(), b[k].numpy()), (net, k)


def test_policy_step_vs_reference(golden):
    g = golden('actor_critic_step.npz')
    ac = make_ac(60, 2, g)
    obs = torch.from_numpy(g['obs']).to(DEV)
    act, v_r, v_c, logp = ac.step(obs, eps=torch.from_numpy(g['eps']).to(DEV))
    np.testing.assert_allclose(act.cpu().numpy(), g['act'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(v_r.cpu().numpy(), g['value_r'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(v_c.cpu().numpy(), g['value_c'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logp.cpu().numpy(), g['logp'], rtol=1e-4, atol=1e-5)
    a_det, _, _, lp_det = ac.step(obs, deterministic=True)
    np.testing.assert_allclose(a_det.cpu().numpy(), g['act_det'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(lp_det.cpu().numpy(), g['logp_det'], rtol=1e-4, atol=1e-5)
    # single-row call (bootstrap path of the adapter)
    a1, vr1, vc1, _ = ac.step(obs[5], deterministic=True)
    np.testing.assert_allclose(vr1.cpu().numpy(), g['value_r'][5], rtol=1e-4, atol=1e-5)
    assert a1.shape == (2,)


@pytest.mark.parametrize('obs_dim,act_dim,N', [(60, 2, 4096), (27, 8, 100), (376, 17, 130), (72, 2, 1),
                                               (5, 1, 64)])
def test_policy_step_vs_oracle_shapes(obs_dim, act_dim, N):
    """BASELINE config shapes (PointGoal1 60/2, Ant 27/8, Humanoid 376/17, CarGoal1 72/2) incl. obs_dim
    not a multiple of 4 or 16, act_dim > 16 (two output tiles), ragged N."""
    torch.manual_seed(obs_dim * 7 + act_dim)
    ref = O.ActorCritic(obs_dim, act_dim)
    ac = make_ac(obs_dim, act_dim)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        getattr(ac, net).load_state_dict(getattr(ref, net).state_dict())
    with torch.no_grad():
        ref.actor.log_std.copy_(torch.linspace(-0.5, 0.3, act_dim))
    ac.actor.load_state_dict(ref.actor.state_dict())
    obs = torch.randn(N, obs_dim)
    eps = torch.randn(N, act_dim)
    act, v_r, v_c, logp = ref.step(obs, eps=eps)
    a2, r2, c2, l2 = ac.step(obs.to(DEV), eps=eps.to(DEV))
    np.testing.assert_allclose(a2.cpu().numpy(), act.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(r2.cpu().numpy(), v_r.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(c2.cpu().numpy(), v_c.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(l2.cpu().numpy(), logp.numpy(), rtol=1e-4, atol=5e-5)


def test_device_noise_is_standard_normal():
    ac = make_ac(60, 2)
    ac.set_seed(7)
    obs = torch.zeros(65536, 60, device=DEV)
    with torch.no_grad():
        a1, _, _, lp = ac.step(obs)
        mean_det, _, _, _ = ac.step(obs, deterministic=True)
        a2, _, _, _ = ac.step(obs)
    e = (a1 - mean_det).double()  # log_std = 0 -> act - mean = eps
    assert abs(float(e.mean())) < 0.02 and abs(float(e.std()) - 1.0) < 0.02
    assert abs(float((e ** 4).mean()) - 3.0) < 0.2  # kurtosis of a normal
    assert not torch.equal(a1, a2)  # the counter advances between calls
    # logp is consistent with the sampled action
    ref_lp = (-0.5 * e ** 2 - 0.9189385332).sum(-1)
    np.testing.assert_allclose(lp.cpu().numpy(), ref_lp.cpu().numpy(), rtol=1e-4, atol=1e-4)


def _update_data(g, dev=DEV):
    a_r, a_c, _ = O.buffer_get(g['buffer/adv_r'], g['buffer/adv_c'])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    d = {'obs': t(O.env_major(g['buffer/obs'])), 'act': t(O.env_major(g['buffer/act'])),
         'logp': t(O.env_major(g['buffer/logp'])),
         'target_value_r': t(O.env_major(g['buffer/target_value_r'])),
         'target_value_c': t(O.env_major(g['buffer/target_value_c'])), 'adv_r': t(a_r), 'adv_c': t(a_c)}
    return d, {k: v.to(dev) for k, v in d.items()}


def test_single_minibatch_step_vs_oracle(golden):
    """One optimiser step of all three networks: losses, clipped gradients' effect, post-Adam params."""
    from omnisafe_amd.update import PPOUpdater

    g = golden('ppolag_epoch.npz')
    cpu, dev = _update_data(g)
    ac = make_ac(60, 2, g, 'init/')
    ref = O.ActorCritic(60, 2)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        getattr(ref, net).load_state_dict({k: v.cpu() for k, v in getattr(ac, net).state_dict().items()})
    lam = 0.37
    idx = torch.from_numpy(g['update/perms'][0][:64].copy())
    lr_, lc_, lp_ = [], [], []
    lr_.append(O.critic_step(ref.reward_critic, ref.reward_critic_optimizer, cpu['obs'][idx],
                             cpu['target_value_r'][idx]))
    lc_.append(O.critic_step(ref.cost_critic, ref.cost_critic_optimizer, cpu['obs'][idx],
                             cpu['target_value_c'][idx]))
    lp, ent, ratio = O.actor_step(ref.actor, ref.actor_optimizer, cpu['obs'][idx], cpu['act'][idx],
                                  cpu['logp'][idx], cpu['adv_r'][idx], cpu['adv_c'][idx], lam)
    up = PPOUpdater(ac, batch_size=64, update_iters=1, target_kl=0.02, kl_early_stop=False)
    up.hp.lr_actor = up.hp.lr_critic = 3e-4
    stats = torch.zeros(16, device=DEV)
    up.minibatch(dev, idx.to(DEV), 64, torch.tensor([lam], device=DEV), stats)
    s = stats.cpu().numpy()
    np.testing.assert_allclose(s[0] + 0.001 * s[5], lr_[0], rtol=1e-4)
    np.testing.assert_allclose(s[1] + 0.001 * s[6], lc_[0], rtol=1e-4)
    np.testing.assert_allclose(s[2], lp, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(s[3], float(ratio.mean()), rtol=1e-5)
    np.testing.assert_allclose(s[4], ent, rtol=1e-6)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in getattr(ac, net).state_dict().items():
            np.testing.assert_allclose(v.cpu().numpy(), getattr(ref, net).state_dict()[k].numpy(),
                                       rtol=1e-4, atol=1e-6, err_msg=f'{net}/{k}')
    assert ac.adam_step.cpu().tolist() == [1, 1, 1]


def test_ppolag_update_vs_reference(golden):
    """Whole PolicyGradient._update (3 passes x 3 minibatches incl. a ragged last one of 32 rows) from
    the reference's initial parameters, buffer contents and recorded permutations."""
    from omnisafe_amd.update import PPOUpdater

    g = golden('ppolag_epoch.npz')
    _, dev = _update_data(g)
    ac = make_ac(60, 2, g, 'init/')
    up = PPOUpdater(ac, batch_size=64, update_iters=3, target_kl=0.02, kl_early_stop=False)
    lam = torch.tensor([float(g['update/lambda_after'])], device=DEV)
    out = up.run(dev, lam, perms=[torch.from_numpy(p.copy()) for p in g['update/perms']],
                 actor_lr=3e-4, critic_lr=3e-4)
    assert out['stop_iter'] == int(g['update/stop_iter'][-1]) and out['steps'] == 9
    summ = PPOUpdater.summarize(out, 0.001, True)
    ps = summ['per_step']
    np.testing.assert_allclose(ps['loss_r'], g['update/loss_r'], rtol=2e-4)
    np.testing.assert_allclose(ps['loss_c'], g['update/loss_c'], rtol=2e-4)
    np.testing.assert_allclose(ps['loss_pi'], g['update/loss_pi'], rtol=2e-3, atol=2e-6)
    np.testing.assert_allclose(ps['ratio_mean'], g['update/ratio_mean'], rtol=1e-5)
    np.testing.assert_allclose(out['kl'], g['update/kl'][-1], rtol=2e-3, atol=1e-7)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in getattr(ac, net).state_dict().items():
            np.testing.assert_allclose(v.cpu().numpy(), g[f'post/{net}/{k}'], rtol=0, atol=5e-6,
                                       err_msg=f'{net}/{k}')


@pytest.mark.parametrize('M,obs_dim,act_dim', [(4096, 60, 2), (5000, 72, 8), (2500, 27, 8)])
def test_large_batch_multiblock_equals_single_pass(M, obs_dim, act_dim):
    """A large minibatch split over up to 64 workgroups + slab reduction == oracle full-batch step: B = 4096
    and a ragged B = 5000 with an 8-D action space on the persistent kernel's partial-gradient mode
    (LDS-resident weights, register accumulators), B = 2500 with unaligned 27-float rows on the per-chunk
    kernel."""
    from omnisafe_amd.update import PPOUpdater

    torch.manual_seed(3)
    ref = O.ActorCritic(obs_dim, act_dim)
    ac = make_ac(obs_dim, act_dim)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        getattr(ac, net).load_state_dict(getattr(ref, net).state_dict())
    cpu = {'obs': torch.randn(M, obs_dim), 'act': torch.randn(M, act_dim), 'logp': -2 + 0.1 * torch.randn(M),
           'target_value_r': torch.randn(M), 'target_value_c': torch.randn(M), 'adv_r': torch.randn(M),
           'adv_c': torch.randn(M)}
    with torch.no_grad():
        d = ref.actor.dist(cpu['obs'])
        cpu['logp'] = d.log_prob(cpu['act']).sum(-1) + 0.3 * torch.randn(M)  # ratios spread around 1
    dev = {k: v.to(DEV) for k, v in cpu.items()}
    lam = 0.5
    l_r = O.critic_step(ref.reward_critic, ref.reward_critic_optimizer, cpu['obs'], cpu['target_value_r'])
    l_c = O.critic_step(ref.cost_critic, ref.cost_critic_optimizer, cpu['obs'], cpu['target_value_c'])
    # (entropy bonus ON: the partial-gradient slabs must not carry the entropy term -- the slab reduction adds it once;
    # until round 3 each of the up to 64 slabs did, invisible with the YAML default entropy_coef = 0)
    l_p, ent, ratio = O.actor_step(ref.actor, ref.actor_optimizer, cpu['obs'], cpu['act'], cpu['logp'],
                                   cpu['adv_r'], cpu['adv_c'], lam, entropy_coef=0.02)
    up = PPOUpdater(ac, batch_size=M, update_iters=1, target_kl=0.02, kl_early_stop=False, max_blocks=64,
                    entropy_coef=0.02)
    up.hp.lr_actor = up.hp.lr_critic = 3e-4
    stats = torch.zeros(16, device=DEV)
    up.minibatch(dev, None, M, torch.tensor([lam], device=DEV), stats)
    s = stats.cpu().numpy()
    np.testing.assert_allclose(s[0] + 0.001 * s[5], l_r, rtol=1e-4)
    np.testing.assert_allclose(s[1] + 0.001 * s[6], l_c, rtol=1e-4)
    np.testing.assert_allclose(s[2], l_p, rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(s[3], float(ratio.mean()), rtol=1e-4)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in getattr(ac, net).state_dict().items():
            np.testing.assert_allclose(v.cpu().numpy(), getattr(ref, net).state_dict()[k].numpy(),
                                       rtol=1e-4, atol=2e-6, err_msg=f'{net}/{k}')


@pytest.mark.parametrize('obs_dim,act_dim,B,critics_only', [(60, 2, 16384, False), (60, 2, 10000, False),
                                                            (72, 17, 6000, False), (60, 2, 16384, True),
                                                            (16, 1, 2048, False)])
def test_balanced_partial_gradients_equal_the_strided_form(obs_dim, act_dim, B, critics_only, monkeypatch):
    """Round 4: the large-batch step's partial gradients with the chunk-tasks of all networks shared evenly by one
    workgroup per compute unit (osa_ppo_part_kernel: contiguous task ranges, ranges that cross a network boundary
    processed in two segments with the weights reloaded) against the strided form (64 workgroups x 4 chunks per
    network): same gradients up to the order of the slab sums.  Shapes: 256 chunks x 3 networks (3 tasks per
    workgroup, workgroups 85 and 170 straddle two networks), a ragged 10 000-row minibatch (157 chunks, 2 tasks per
    workgroup), two output tiles, critics only (2 networks in the mask), fewer tasks than compute units."""
    from omnisafe_amd.update import PPOUpdater

    M = B
    torch.manual_seed(11)
    data = {'obs': torch.randn(M, obs_dim, device=DEV), 'act': torch.randn(M, act_dim, device=DEV),
            'target_value_r': torch.randn(M, device=DEV), 'target_value_c': torch.randn(M, device=DEV),
            'adv_r': torch.randn(M, device=DEV), 'adv_c': torch.randn(M, device=DEV)}
    perm = torch.randperm(M, device=DEV)
    res = []
    for balanced in ('1', '0'):
        monkeypatch.setenv('OSA_LARGE_BATCH_BALANCED', balanced)
        torch.manual_seed(5)
        ac = make_ac(obs_dim, act_dim)
        if 'logp' not in data:
            _, _, _, lp = ac.step(data['obs'], eps=(data['act'] * 0))
            data['logp'] = lp + 0.2 * torch.randn(M, device=DEV)
        up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False, entropy_coef=0.01,
                        update_actor=not critics_only)
        up.hp.lr_actor, up.hp.lr_critic = 3e-4, 1e-3
        p0 = ac.params.clone()
        lam = torch.tensor([0.3], device=DEV)
        stats = torch.zeros(2, 16, device=DEV)
        for k in range(2):
            up.minibatch(data, perm, B, lam, stats[k])
        torch.cuda.synchronize()
        up.check_reduce_sync()
        res.append((ac.params.clone(), ac.adam_m.clone(), stats.clone(), p0))
    (pa, ma, sa, p0), (pb, mb_, sb, _) = res
    assert float((pa - p0).abs().max()) > 2e-4  # two Adam steps happened
    if critics_only:
        assert torch.equal(pa[0], p0[0])
    # first moments = 0.1 g_1 (0.9) + 0.1 g_2: the gradients themselves, to summation-order accuracy
    scale = float(mb_.abs().max())
    assert float((ma - mb_).abs().max()) <= 2e-6 * scale + 1e-9, (float((ma - mb_).abs().max()), scale)
    # parameters: Adam's first steps are lr g / (|g| + 1e-8) -- the few elements whose gradient is itself of the order
    # of eps turn a 1e-9 difference of the slab sums into a visible fraction of ONE lr step; everything else agrees
    # to float32 rounding
    d = (pa - pb).abs()
    assert float((d > 1e-6).float().mean()) < 1e-3 and float(d.max()) < 2e-3, (float((d > 1e-6).float().mean()), float(d.max()))
    np.testing.assert_allclose(sa[:, :10].cpu().numpy(), sb[:, :10].cpu().numpy(), rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize('obs_dim,act_dim,M,B', [(60, 2, 4096, 64), (27, 8, 1000, 64), (72, 2, 640, 32),
                                                (90, 17, 512, 64), (5, 1, 130, 64), (60, 2, 1000, 128),
                                                (72, 2, 700, 200), (72, 2, 4096, 128), (27, 8, 2048, 128),
                                                (60, 2, 1100, 512), (60, 2, 2100, 1000),
                                                # wide observations: osa_ppo_wide_pass (W1 streamed from L2)
                                                (376, 17, 1024, 64), (376, 17, 200, 64), (100, 3, 300, 64),
                                                (200, 20, 256, 32), (512, 32, 192, 64), (97, 1, 130, 48)])
def test_persistent_pass_equals_per_minibatch_launches(obs_dim, act_dim, M, B, monkeypatch):
    """osa_ppo_pass (one persistent launch per pass, weights in LDS, Adam moments in registers) vs
    osa_ppo_minibatch (one launch per optimiser step): same parameters, moments and statistics after
    two passes, including ragged last minibatches and every (KB, OT) template instance family.  Wide
    observations: both persistent kernels (first layer split over cooperating CUs = the default, and the
    one-CU kernel behind OSA_WIDE_SPLIT=0) against the per-step launches."""
    from omnisafe_amd import _lib
    from omnisafe_amd.update import PPOUpdater

    narrow = bool(_lib.load().osa_ppo_pass_supported(obs_dim, act_dim, 64))
    first = ('persistent-chunked' if B > 64 else 'persistent') if narrow else 'persistent-wide-split'
    variants = [(True, '1', first), (False, '1', 'per-step')]
    if not narrow:  # '1' = one XCC per network (L2 hand-offs); 'spread' = all XCCs, uncached exchange buffer
        variants.insert(1, (True, 'spread', 'persistent-wide-split'))
        variants.insert(2, (True, '0', 'persistent-wide'))
    elif B > 64:  # the minibatch's 64-row chunks on cooperating workgroups (default) / walked by one workgroup
        variants.insert(1, (True, 'nochunk', 'persistent' if B <= 512 else 'per-step'))  # (one workgroup: up to 512)

    torch.manual_seed(obs_dim + act_dim)
    data = {'obs': torch.randn(M, obs_dim, device=DEV), 'act': torch.randn(M, act_dim, device=DEV),
            'target_value_r': torch.randn(M, device=DEV), 'target_value_c': torch.randn(M, device=DEV),
            'adv_r': torch.randn(M, device=DEV), 'adv_c': torch.randn(M, device=DEV)}
    acs, outs, paths = [], [], []
    perms = [torch.randperm(M), torch.randperm(M)]
    for persistent, split, _ in variants:
        monkeypatch.setenv('OSA_WIDE_SPLIT', split)
        monkeypatch.setenv('OSA_CHUNKED_PASS', '0' if split == 'nochunk' else '1')
        torch.manual_seed(99)
        ac = make_ac(obs_dim, act_dim)
        if 'logp' not in data:
            _, _, _, lp = ac.step(data['obs'], eps=(data['act'] * 0))  # logp of mean action
            data['logp'] = lp + 0.2 * torch.randn(M, device=DEV)
        up = PPOUpdater(ac, batch_size=B, update_iters=2, target_kl=0.02, kl_early_stop=False,
                        entropy_coef=0.01, persistent=persistent)
        lam = torch.tensor([0.3], device=DEV)
        outs.append(up.run(data, lam, perms=perms, actor_lr=3e-4, critic_lr=1e-3))
        paths.append(up.last_path)
        acs.append(ac)
    assert all(o['steps'] == 2 * ((M + B - 1) // B) for o in outs)
    assert paths == [v[2] for v in variants]
    assert narrow == (obs_dim <= 96 and not (obs_dim > 80 and act_dim > 16))
    # B <= 64: the same operation order except for the 1-2-output layers, which the pass kernel evaluates
    # on the VALU (16-term partial dot products per lane group) and the per-step kernels on MFMA tiles:
    # float32 summation-order differences of ~1e-7.  B > 64: the pass kernel accumulates the 64-row chunks
    # in registers, the per-step path reduces per-workgroup slabs -- Adam's m/sqrt(v) amplifies that
    # order difference to ~1e-6 in the parameters.
    # wide inputs: 376-term dot products; Adam's m / sqrt(v) turns ~1e-7 gradient differences into up to ~2e-6
    # in single parameters with small v (measured: 1 element of 1e5 at 1.6e-6)
    atol = (5e-7 if narrow else 5e-6) if B <= 64 else 5e-6
    for k in range(len(variants) - 1):  # every persistent variant against the per-step launches
        assert acs[k].adam_step.cpu().tolist() == acs[-1].adam_step.cpu().tolist()
        for name in ('params', 'adam_m', 'adam_v'):
            a, b = getattr(acs[k], name).cpu().numpy(), getattr(acs[-1], name).cpu().numpy()
            if variants[k][2] not in ('persistent-wide-split', 'persistent-chunked'):
                np.testing.assert_allclose(a, b, rtol=1e-5, atol=atol, err_msg=f'{variants[k][2]} {name}')
                continue
            # The split kernel sums the layer-1 pre-activation as C partial sums (the other persistent kernels keep
            # the per-step kernels' order and agree to 1e-8): every gradient differs by ~1e-7 relative; the chunked
            # pass sums per-chunk gradients where one workgroup accumulates all chunks in its MFMA accumulators.  Adam's
            # FIRST step is lr * g / (|g| + 1e-8): for the handful of elements whose first gradient is within
            # ~1e-8 of zero (expected: ~1e-5 of all elements) that noise moves the update by a visible
            # fraction of lr, and the moments follow at the 1e-4 relative level.  Everything else obeys the usual tolerance.
            bad = np.abs(a - b) > 5e-6 + 1e-5 * np.abs(b)
            assert bad.sum() <= 4, (name, int(bad.sum()))
            if bad.any():
                lim = 2.5e-4 if name == 'params' else 2e-3 * np.abs(b[bad]).max()
                assert np.abs(a - b)[bad].max() <= lim, (name, float(np.abs(a - b)[bad].max()))
        s0, s1 = outs[k]['stats'].cpu().numpy(), outs[-1]['stats'].cpu().numpy()
        np.testing.assert_allclose(s0[:, :10], s1[:, :10], rtol=1e-5, atol=2e-7)
        np.testing.assert_allclose(outs[k]['kl'], outs[-1]['kl'], rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize('W,M,B,use_graph,coop', [(2, 512, 64, False, False), (4, 300, 64, True, False),
                                                  (3, 256, 128, True, False), (2, 512, 64, False, True),
                                                  (4, 300, 64, False, True), (3, 256, 128, False, True),
                                                  (8, 1024, 64, False, True), (1, 200, 64, False, True),
                                                  (6, 320, 64, False, True), (16, 128, 64, False, True),
                                                  (2, 384, 64, False, 'wrong-placement'),
                                                  (11, 128, 64, False, True), (8, 256, 64, False, True)])
def test_replicated_data_parallel_step_equals_allreduce_semantics(W, M, B, use_graph, coop, monkeypatch):
    """osa_ppo_dp_step (every rank computes the whole global step on the all-gathered rollout: W
    workgroups per network -> average of the locally clipped gradients -> Adam) vs the reference's
    data-parallel semantics emulated rank by rank with the per-step kernels (gradient + local clip per
    rank, average, Adam): policy_gradient.py:437-442, distributed.py:193-198."""
    import ctypes as C

    from omnisafe_amd import _lib
    from omnisafe_amd import update as U
    from omnisafe_amd.update import PPOUpdater

    if coop == 'wrong-placement':  # test hook: the one-XCC protocol on the spread grid -> the kernel's placement
        # check trips before anything is modified, the updater repeats the pass spread over the XCCs
        monkeypatch.setenv('OSA_DEBUG_PLACEMENT', 'wrong')
        monkeypatch.setenv('OSA_DP_XCH', 'local')
        monkeypatch.setitem(U._PLACEMENT, 'local_ok', None)
    torch.manual_seed(W * 100 + M)
    obs_dim, act_dim = 60, 2
    data_all = {'obs': torch.randn(W * M, obs_dim, device=DEV), 'act': torch.randn(W * M, act_dim, device=DEV),
                'target_value_r': torch.randn(W * M, device=DEV), 'target_value_c': torch.randn(W * M, device=DEV),
                'adv_r': torch.randn(W * M, device=DEV), 'adv_c': torch.randn(W * M, device=DEV)}
    perms = [torch.stack([torch.randperm(M) for _ in range(W)]).to(DEV) for _ in range(3)]
    lam = torch.tensor([0.4], device=DEV)
    nmb = (M + B - 1) // B
    results = []
    for mode in ('replicated', 'emulated'):
        torch.manual_seed(5)
        ac = make_ac(obs_dim, act_dim)
        if 'logp' not in data_all:
            _, _, _, lp = ac.step(data_all['obs'], eps=data_all['act'] * 0)
            data_all['logp'] = lp + 0.2 * torch.randn(W * M, device=DEV)
        up = PPOUpdater(ac, batch_size=B, update_iters=3, target_kl=0.02, kl_early_stop=False, entropy_coef=0.01)
        up.hp.lr_actor, up.hp.lr_critic = 3e-4, 1e-3
        stats = torch.zeros(3 * nmb, 16, device=DEV)
        if mode == 'replicated':
            for i in range(3):
                up.run_pass_replicated(data_all, M, W, lam, stats[i * nmb:(i + 1) * nmb], perms_all=perms[i],
                                       use_graph=use_graph, coop=bool(coop))
            if coop:  # the cooperative persistent launch ran (no silent fallback) and every peer arrived
                assert up._dp.get('coop_passes') == 3
                up.check_dp_sync()
            if coop == 'wrong-placement':
                assert U._PLACEMENT['local_ok'] is False and up._dp['local'] is False
            if use_graph:
                assert up._dp.get('graph') is not None and not up._dp.get('graph_failed', False)
        else:
            lib = _lib.load()
            row = torch.zeros(16, device=DEV)
            for i in range(3):
                for k in range(nmb):
                    acc = torch.zeros_like(ac.grads)
                    for r in range(W):
                        idx = (perms[i][r, k * B:(k + 1) * B] + r * M).contiguous()
                        _lib.check(lib.osa_ppo_minibatch(
                            obs_dim, act_dim, 64, _lib.ptr(ac.params), _lib.ptr(ac.adam_m), _lib.ptr(ac.adam_v),
                            _lib.ptr(ac.adam_step), _lib.ptr(ac.grads), _lib.ptr(data_all['obs']), obs_dim,
                            _lib.ptr(data_all['act']), act_dim, _lib.ptr(data_all['logp']),
                            _lib.ptr(data_all['target_value_r']), _lib.ptr(data_all['target_value_c']),
                            _lib.ptr(data_all['adv_r']), _lib.ptr(data_all['adv_c']), _lib.ptr(idx), idx.numel(),
                            _lib.ptr(lam), C.byref(up.hp), 0, 1, 7, 4, _lib.ptr(up._ws), _lib.ptr(row),
                            _lib.stream_ptr()))
                        acc += ac.grads
                    ac.grads.copy_(acc / W)
                    _lib.check(lib.osa_adam_apply(obs_dim, act_dim, 64, _lib.ptr(ac.params), _lib.ptr(ac.adam_m),
                                                  _lib.ptr(ac.adam_v), _lib.ptr(ac.adam_step), _lib.ptr(ac.grads),
                                                  C.byref(up.hp), 7, _lib.stream_ptr()))
        results.append((ac, stats.cpu().numpy()))
    a0, a1 = results[0][0], results[1][0]
    assert a0.adam_step.cpu().tolist() == a1.adam_step.cpu().tolist() == [3 * nmb] * 3
    for name in ('params', 'adam_m', 'adam_v'):
        np.testing.assert_allclose(getattr(a0, name).cpu().numpy(), getattr(a1, name).cpu().numpy(), rtol=1e-5,
                                   atol=5e-6, err_msg=name)
    assert np.isfinite(results[0][1][:, :10]).all() and (results[0][1][:, 3] > 0).all()


@pytest.mark.parametrize('W,M,obs_dim,act_dim,mode_dp', [
    (2, 384, 376, 17, 'place'), (4, 256, 376, 17, 'place'), (8, 192, 376, 17, 'place'), (3, 200, 128, 6, 'place'),
    (1, 128, 376, 17, 'place'), (5, 150, 200, 1, 'place'), (8, 192, 376, 17, 'spread'), (3, 200, 128, 6, 'spread'),
    (2, 384, 376, 17, 'wrong-placement')])
def test_wide_split_data_parallel_pass_equals_allreduce_semantics(W, M, obs_dim, act_dim, mode_dp, monkeypatch):
    """osa_ppo_split_dp_pass (BASELINE config 4 under world_size > 1: W virtual ranks x 3 networks x (leader +
    helpers) in ONE cooperative launch; the owners of the same parameters average their locally clipped shares)
    vs the reference's data-parallel semantics emulated rank by rank with the per-step kernels (gradient + local
    clip per rank, average, Adam): policy_gradient.py:437-442, 478-483, 519-524, distributed.py:167-198.  Includes a
    ragged last minibatch (M % 64 != 0) and a max_grad_norm small enough that the clip is ACTIVE on some ranks."""
    import ctypes as C

    from omnisafe_amd import _lib
    from omnisafe_amd import update as U
    from omnisafe_amd.update import PPOUpdater

    # 'place': the W owners of the same parameters on one XCC (default); 'spread': rank-major, uncached exchange;
    # 'wrong-placement': test hook -- the placed protocol on the rank-major grid: the kernel's placement check trips
    # before anything is modified and the updater repeats the pass spread
    monkeypatch.setenv('OSA_WIDE_DP', 'spread' if mode_dp == 'spread' else 'place')
    monkeypatch.setitem(U._PLACEMENT, 'local_ok', None)
    if mode_dp == 'wrong-placement':
        monkeypatch.setenv('OSA_DEBUG_PLACEMENT', 'wrong')
    torch.manual_seed(W * 1000 + M)
    B = 64
    data_all = {'obs': torch.randn(W * M, obs_dim, device=DEV), 'act': torch.randn(W * M, act_dim, device=DEV),
                'target_value_r': torch.randn(W * M, device=DEV) * 3, 'target_value_c': torch.randn(W * M, device=DEV),
                'adv_r': torch.randn(W * M, device=DEV), 'adv_c': torch.randn(W * M, device=DEV)}
    perms = [torch.stack([torch.randperm(M) for _ in range(W)]).to(DEV) for _ in range(2)]
    lam = torch.tensor([0.4], device=DEV)
    nmb = (M + B - 1) // B
    results = []
    for mode in ('replicated', 'emulated'):
        torch.manual_seed(5)
        ac = make_ac(obs_dim, act_dim)
        if 'logp' not in data_all:
            _, _, _, lp = ac.step(data_all['obs'], eps=data_all['act'] * 0)
            data_all['logp'] = lp + 0.2 * torch.randn(W * M, device=DEV)
        # max_grad_norm 1.5: the reward critic's gradient (targets x 3) exceeds it on most steps, the actor's on few
        up = PPOUpdater(ac, batch_size=B, update_iters=2, target_kl=0.02, kl_early_stop=False, entropy_coef=0.01,
                        max_grad_norm=1.5)
        up.hp.lr_actor, up.hp.lr_critic = 3e-4, 1e-3
        stats = torch.zeros(2 * nmb, 16, device=DEV)
        if mode == 'replicated':
            assert up._wide_dp_fits(W)
            up._repl_wide = True
            for i in range(2):
                up.run_pass_replicated(data_all, M, W, lam, stats[i * nmb:(i + 1) * nmb], perms_all=perms[i])
            up.check_wide_dp_sync()
            assert up._dp['wide_place'] is (mode_dp == 'place')
            if mode_dp == 'wrong-placement':
                assert U._PLACEMENT['local_ok'] is False
        else:
            lib = _lib.load()
            row = torch.zeros(16, device=DEV)
            norms = []
            for i in range(2):
                for k in range(nmb):
                    acc = torch.zeros_like(ac.grads)
                    for r in range(W):
                        idx = (perms[i][r, k * B:(k + 1) * B] + r * M).contiguous()
                        _lib.check(lib.osa_ppo_minibatch(
                            obs_dim, act_dim, 64, _lib.ptr(ac.params), _lib.ptr(ac.adam_m), _lib.ptr(ac.adam_v),
                            _lib.ptr(ac.adam_step), _lib.ptr(ac.grads), _lib.ptr(data_all['obs']), obs_dim,
                            _lib.ptr(data_all['act']), act_dim, _lib.ptr(data_all['logp']),
                            _lib.ptr(data_all['target_value_r']), _lib.ptr(data_all['target_value_c']),
                            _lib.ptr(data_all['adv_r']), _lib.ptr(data_all['adv_c']), _lib.ptr(idx), idx.numel(),
                            _lib.ptr(lam), C.byref(up.hp), 0, 1, 7, 4, _lib.ptr(up._ws), _lib.ptr(row),
                            _lib.stream_ptr()))
                        acc += ac.grads
                        norms.append(row[7:10].cpu().numpy().copy())
                    ac.grads.copy_(acc / W)
                    _lib.check(lib.osa_adam_apply(obs_dim, act_dim, 64, _lib.ptr(ac.params), _lib.ptr(ac.adam_m),
                                                  _lib.ptr(ac.adam_v), _lib.ptr(ac.adam_step), _lib.ptr(ac.grads),
                                                  C.byref(up.hp), 7, _lib.stream_ptr()))
            norms = np.asarray(norms)
            assert (norms[:, 1] > 1.5).mean() > 0.5  # the clip really is active (reward critic)
        results.append((ac, stats.cpu().numpy()))
    a0, a1 = results[0][0], results[1][0]
    assert a0.adam_step.cpu().tolist() == a1.adam_step.cpu().tolist() == [2 * nmb] * 3
    for name in ('params', 'adam_m', 'adam_v'):
        a, b = getattr(a0, name).cpu().numpy(), getattr(a1, name).cpu().numpy()
        # (tolerance of the single-rank split pass against the per-step kernels, test_persistent_pass_equals_...:
        # the layer-1 pre-activation is a sum of partial sums, every gradient differs by ~1e-7 relative, and Adam's
        # first step amplifies that for the few elements whose first gradient is within ~1e-8 of zero)
        bad = np.abs(a - b) > 5e-6 + 1e-5 * np.abs(b)
        assert bad.sum() <= 6, (name, int(bad.sum()))
        if bad.any():
            lim = 2.5e-4 if name == 'params' else 2e-3 * np.abs(b[bad]).max()
            assert np.abs(a - b)[bad].max() <= lim, (name, float(np.abs(a - b)[bad].max()))
    st = results[0][1]
    assert np.isfinite(st[:, :10]).all() and (st[:, 3] > 0).all() and (st[:, 7:10] > 0).all()


@pytest.mark.parametrize('case', ['wide place W8', 'wide spread W8', 'wide place W3 128/6', 'narrow placed W8',
                                  'chunked W8 27/8', 'single wide split local', 'single chunked 72/2 B128'])
def test_cooperative_passes_reproduce_themselves_bit_for_bit(case, monkeypatch):
    """Race hunt (tools/dp_stress.py): the same cooperative pass executed 500 times from the same initial state, other
    kernels of varying length in between -- every execution must reproduce the first one bit for bit (the workgroups
    sum in a fixed order; nothing in the arithmetic depends on timing) with the sticky words at 0.  Round 3 found
    stale dz1 reads in the split pass this way (about 1 pass in 1000 at 8 virtual ranks: the helper polled a flag and
    requested the data right behind it in the same memory round trip; loads return in issue order but are not
    performed in it) -- the intermittent failure of test_wide_split_data_parallel_pass_... inside full-suite runs."""
    import importlib.util
    import os

    from omnisafe_amd import update as U

    spec = importlib.util.spec_from_file_location(
        'dp_stress', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'dp_stress.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for k in ('OSA_WIDE_DP', 'OSA_WIDE_SPLIT'):
        monkeypatch.setenv(k, os.environ.get(k, ''))  # (run_case sets them: restored after the test)
        monkeypatch.delenv(k)
    monkeypatch.setitem(U._PLACEMENT, 'local_ok', None)
    rec = mod.run_case(next(c for c in mod.CASES if c[0] == case), 500)
    assert rec['mismatching_runs'] == 0 and not rec['messages'], rec


@pytest.mark.parametrize('W,M,B,obs_dim,act_dim,chunked', [
    (2, 512, 128, 27, 8, True), (4, 384, 128, 27, 8, True), (8, 256, 128, 27, 8, True), (3, 300, 128, 72, 2, True),
    (2, 640, 256, 60, 2, True), (1, 256, 128, 27, 8, True), (4, 384, 128, 27, 8, False)])
def test_chunked_data_parallel_pass_equals_allreduce_semantics(W, M, B, obs_dim, act_dim, chunked, monkeypatch):
    """osa_ppo_dp_chunked_pass (the batch-128 critic / actor passes of BASELINE configs 3 and 5 under world_size > 1:
    W ranks x ceil(B / 64) chunk workgroups per network, rank sum -> rank clip -> average over ranks) vs the
    reference's data-parallel semantics emulated rank by rank with the per-step kernels (B-row gradient + local
    clip per rank, average, Adam): natural_pg.py:205-223, policy_gradient.py:437-442, distributed.py:167-198.
    max_grad_norm is small enough that the clip is active on most steps of the reward critic.  `chunked = False`:
    the same through one workgroup per rank walking the chunks (OSA_CHUNKED_PASS=0)."""
    import ctypes as C

    from omnisafe_amd import _lib
    from omnisafe_amd.update import PPOUpdater

    monkeypatch.setenv('OSA_CHUNKED_PASS', '1' if chunked else '0')
    torch.manual_seed(W * 1000 + M + B)
    ld = (obs_dim + 3) // 4 * 4  # 16-byte aligned rows (update.py pads once per update; here by construction)
    obs = torch.randn(W * M, ld, device=DEV)[:, :obs_dim]
    data_all = {'obs': obs, 'act': torch.randn(W * M, act_dim, device=DEV),
                'target_value_r': torch.randn(W * M, device=DEV) * 3, 'target_value_c': torch.randn(W * M, device=DEV),
                'adv_r': torch.randn(W * M, device=DEV), 'adv_c': torch.randn(W * M, device=DEV)}
    perms = [torch.stack([torch.randperm(M) for _ in range(W)]).to(DEV) for _ in range(2)]
    lam = torch.tensor([0.4], device=DEV)
    nmb = (M + B - 1) // B
    results = []
    for mode in ('replicated', 'emulated'):
        torch.manual_seed(5)
        ac = make_ac(obs_dim, act_dim)
        if 'logp' not in data_all:
            _, _, _, lp = ac.step(data_all['obs'].contiguous(), eps=data_all['act'] * 0)
            data_all['logp'] = lp + 0.2 * torch.randn(W * M, device=DEV)
        up = PPOUpdater(ac, batch_size=B, update_iters=2, target_kl=0.02, kl_early_stop=False, entropy_coef=0.01,
                        max_grad_norm=1.5)
        up.hp.lr_actor, up.hp.lr_critic = 3e-4, 1e-3
        stats = torch.zeros(2 * nmb, 16, device=DEV)
        if mode == 'replicated':
            for i in range(2):
                up.run_pass_replicated(data_all, M, W, lam, stats[i * nmb:(i + 1) * nmb], perms_all=perms[i],
                                       use_graph=False, coop=True)
            assert up._dp.get('coop_passes') == 2 and up._dp['chunked'] is chunked
            up.check_dp_sync()
        else:
            lib = _lib.load()
            row = torch.zeros(16, device=DEV)
            clipped = 0
            for i in range(2):
                for k in range(nmb):
                    acc = torch.zeros_like(ac.grads)
                    for r in range(W):
                        idx = (perms[i][r, k * B:(k + 1) * B] + r * M).contiguous()
                        _lib.check(lib.osa_ppo_minibatch(
                            obs_dim, act_dim, 64, _lib.ptr(ac.params), _lib.ptr(ac.adam_m), _lib.ptr(ac.adam_v),
                            _lib.ptr(ac.adam_step), _lib.ptr(ac.grads), _lib.ptr(data_all['obs']), ld,
                            _lib.ptr(data_all['act']), act_dim, _lib.ptr(data_all['logp']),
                            _lib.ptr(data_all['target_value_r']), _lib.ptr(data_all['target_value_c']),
                            _lib.ptr(data_all['adv_r']), _lib.ptr(data_all['adv_c']), _lib.ptr(idx), idx.numel(),
                            _lib.ptr(lam), C.byref(up.hp), 0, 1, 7, 8, _lib.ptr(up._ws), _lib.ptr(row),
                            _lib.stream_ptr()))
                        acc += ac.grads
                        clipped += int(float(row[8]) > 1.5)
                    ac.grads.copy_(acc / W)
                    _lib.check(lib.osa_adam_apply(obs_dim, act_dim, 64, _lib.ptr(ac.params), _lib.ptr(ac.adam_m),
                                                  _lib.ptr(ac.adam_v), _lib.ptr(ac.adam_step), _lib.ptr(ac.grads),
                                                  C.byref(up.hp), 7, _lib.stream_ptr()))
            assert clipped >= nmb * W  # the clip really is active (reward critic: at least half of its steps)
        results.append((ac, stats.cpu().numpy()))
    a0, a1 = results[0][0], results[1][0]
    assert a0.adam_step.cpu().tolist() == a1.adam_step.cpu().tolist() == [2 * nmb] * 3
    for name in ('params', 'adam_m', 'adam_v'):
        a, b = getattr(a0, name).cpu().numpy(), getattr(a1, name).cpu().numpy()
        bad = np.abs(a - b) > 5e-6 + 1e-5 * np.abs(b)  # (as the single-rank chunked pass: summation order of the chunks)
        assert bad.sum() <= 6, (name, int(bad.sum()))
        if bad.any():
            lim = 2.5e-4 if name == 'params' else 2e-3 * np.abs(b[bad]).max()
            assert np.abs(a - b)[bad].max() <= lim, (name, float(np.abs(a - b)[bad].max()))
    st = results[0][1]
    assert np.isfinite(st[:, :10]).all() and (st[:, 3] > 0).all() and (st[:, 7:10] > 0).all()


@pytest.mark.parametrize('activation,width', [('relu', 64), ('sigmoid', 64), ('softplus', 64), ('identity', 64),
                                              ('tanh', 64), ('tanh', 128), ('relu', 256), ('tanh', 256),
                                              ('relu', 32), ('softplus', 128)])
def test_hidden_activations_and_widths_vs_oracle(activation, width):
    """model_cfgs.*.activation (reference utils/model.py:47-70: identity / relu / sigmoid / softplus / tanh): the
    per-step kernel family takes the activation code in bits 16-19 of the `hidden` ABI word and hidden_sizes [H, H]
    with H in 32 / 64 / 128 / 256 (256: 32-sample chunks, the four [H][36] tiles fill the LDS); the persistent
    passes are 64-wide tanh only and decline.  Against the oracle's torch modules built alike: policy step,
    one whole PolicyGradient._update (2 passes x 4 minibatches incl. a ragged one, injected permutations), the
    full-batch KL, and the Fisher-vector product (double backward in the oracle, JVP + VJP here)."""
    from omnisafe_amd.trust_region import TrustRegionSolver
    from omnisafe_amd.update import PPOUpdater

    torch.manual_seed(7)
    obs_dim, act_dim, M, B = 24, 3, 230, 64
    ref = O.ActorCritic(obs_dim, act_dim, hidden=(width, width), activation=activation)
    ac = make_ac(obs_dim, act_dim, activation=activation, width=width)
    assert ac.hidden == width | ({'tanh': 0, 'relu': 1, 'sigmoid': 2, 'softplus': 3, 'identity': 4}[activation] << 16)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        getattr(ac, net).load_state_dict(getattr(ref, net).state_dict())
    cpu = {'obs': torch.randn(M, obs_dim) * 1.5, 'act': torch.randn(M, act_dim),
           'target_value_r': torch.randn(M), 'target_value_c': torch.randn(M), 'adv_r': torch.randn(M),
           'adv_c': torch.randn(M)}
    # ---- policy step
    eps = torch.randn(M, act_dim)
    a_ref, vr_ref, vc_ref, lp_ref = ref.step(cpu['obs'], eps=eps)
    a, vr, vc, lp = ac.step(cpu['obs'].to(DEV), eps=eps.to(DEV))
    for got, want in ((a, a_ref), (vr, vr_ref), (vc, vc_ref), (lp, lp_ref)):
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-5)
    with torch.no_grad():
        d = ref.actor.dist(cpu['obs'])
        cpu['logp'] = d.log_prob(cpu['act']).sum(-1) + 0.3 * torch.randn(M)
    dev = {k: v.to(DEV) for k, v in cpu.items()}
    # ---- Fisher-vector product at the initial parameters
    s = TrustRegionSolver(ac, cg_iters=5, cg_damping=0.1)
    s.begin(dev['obs'])
    v = ac.actor.pad(torch.randn(ac.actor.num_params))
    Fv = ac.actor.unpad(s.fvp(v)).cpu().numpy()
    Fv_ref = O.fvp(ref.actor, cpu['obs'], ac.actor.unpad(v).cpu(), cg_damping=0.1).numpy()
    assert np.linalg.norm(Fv - Fv_ref) / np.linalg.norm(Fv_ref) < 2e-3
    # ---- one whole update
    perms = [torch.randperm(M) for _ in range(2)]
    lam = 0.4
    ref_out = O.ppolag_update(ref, cpu, lam, perms, batch_size=B, update_iters=2, kl_early_stop=False)
    up = PPOUpdater(ac, batch_size=B, update_iters=2, target_kl=0.02, kl_early_stop=False)
    out = up.run(dev, torch.tensor([lam], device=DEV), perms=perms, actor_lr=3e-4, critic_lr=3e-4)
    assert up.last_path == ('persistent' if (activation == 'tanh' and width == 64) else 'per-step')
    assert out['steps'] == 8
    for net in ('actor', 'reward_critic', 'cost_critic'):
        want = getattr(ref, net).state_dict()
        for k, t in getattr(ac, net).state_dict().items():
            np.testing.assert_allclose(t.cpu().numpy(), want[k].numpy(), rtol=0, atol=6e-6, err_msg=f'{net}/{k}')
    np.testing.assert_allclose(out['kl'], ref_out['kl'], rtol=5e-3, atol=1e-7)


def test_full_size_pass_properties():
    """BASELINE config 2 at full size (M = 65 536 rows, 1024 optimiser steps of 64 rows per pass): size-
    independent properties of the persistent pass -- run-to-run bit-determinism, the Adam step counters,
    zero padding staying zero, finite statistics, and permutation equivariance of a pass with the learning
    rates at 0 (the statistics of step k then depend only on the rows of step k)."""
    from omnisafe_amd.update import PPOUpdater

    torch.manual_seed(11)
    M, B, obs_dim, act_dim = 65536, 64, 60, 2
    data = {'obs': torch.randn(M, obs_dim, device=DEV), 'act': torch.randn(M, act_dim, device=DEV),
            'target_value_r': torch.randn(M, device=DEV), 'target_value_c': torch.randn(M, device=DEV),
            'adv_r': torch.randn(M, device=DEV), 'adv_c': torch.randn(M, device=DEV)}
    perm = torch.randperm(M)
    lam = torch.tensor([0.2], device=DEV)
    outs = []
    for rep in range(2):
        torch.manual_seed(5)
        ac = make_ac(obs_dim, act_dim)
        if 'logp' not in data:
            _, _, _, lp = ac.step(data['obs'], eps=data['act'] * 0)
            data['logp'] = lp + 0.2 * torch.randn(M, device=DEV)
        up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False)
        out = up.run(data, lam, perms=[perm], actor_lr=3e-4, critic_lr=3e-4)
        outs.append((ac.params.clone(), ac.adam_m.clone(), ac.adam_v.clone(), out['stats'][:, :10].clone(), ac))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][3], outs[1][3])
    ac = outs[0][4]
    assert ac.adam_step.cpu().tolist() == [M // B] * 3
    assert bool(torch.isfinite(outs[0][0]).all()) and bool(torch.isfinite(outs[0][3]).all())
    pad = torch.ones(ac.layout.P, dtype=torch.bool, device=DEV)
    pad[ac.actor._flat_index] = False
    assert float(ac.params[0][pad].abs().max()) == 0.0 and float(ac.adam_v[0][pad].abs().max()) == 0.0
    # learning rate 0: parameters fixed, so step k's statistics are a function of its 64 rows only --
    # reversing the order of the minibatches reverses the rows of the statistics
    stats = []
    for p in (perm, perm.reshape(M // B, B).flip(0).reshape(-1)):
        torch.manual_seed(5)
        ac0 = make_ac(obs_dim, act_dim)
        up = PPOUpdater(ac0, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False)
        stats.append(up.run(data, lam, perms=[p], actor_lr=0.0, critic_lr=0.0)['stats'][:, :5].clone())
    assert torch.equal(stats[0], stats[1].flip(0))


def test_full_size_pass_vs_oracle():
    """The thing bench.py times, against the oracle: ONE full-size pass of BASELINE config 2 (M = 65 536
    rows, batch 64 -> 1024 CHAINED optimiser steps of all three networks in one persistent launch) vs
    np_oracle.ppolag_update (pinned to the reference's `_update()`), same data, same permutation, same
    initial parameters.

    Drift tolerance: both sides compute in float32 but sum in different orders (CPU sgemm vs MFMA tiles), so
    every step's gradient differs by ~1e-7 relative; Adam turns that into parameter differences of up to
    ~lr * 1e-3 per step where sqrt(v) is small, and 1024 dependent steps accumulate them.  Measured on
    MI355X: max |theta - theta_oracle| = 1.0e-7 (actor), 1.3e-7 / 2.1e-7 (critics) while the parameters move
    by 0.02.  Required: <= 2e-6 for every network (10x the measurement), the per-step losses within rtol 2e-3
    for >= 99.5 % of the 1024 steps and in the mean within 1e-4 relative, the final KL within 2 %."""
    import np_oracle as O
    from omnisafe_amd.update import PPOUpdater

    torch.manual_seed(17)
    M, B, obs_dim, act_dim = 65536, 64, 60, 2
    ac = make_ac(obs_dim, act_dim)
    obs = torch.randn(M, obs_dim, device=DEV).clamp_(-5, 5)
    act, _, _, logp = ac.step(obs)  # behaviour policy = initial policy: ratios start at 1
    data = {'obs': obs, 'act': act.clone(), 'logp': logp.clone(),
            'target_value_r': torch.randn(M, device=DEV), 'target_value_c': torch.rand(M, device=DEV),
            'adv_r': torch.randn(M, device=DEV), 'adv_c': torch.randn(M, device=DEV)}
    perm = torch.randperm(M)
    lam = 0.35
    ref = O.ActorCritic(obs_dim, act_dim, actor_lr=3e-4, critic_lr=3e-4)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        getattr(ref, net).load_state_dict({k: v.cpu() for k, v in getattr(ac, net).state_dict().items()})
    init = {k: v.cpu().clone() for k, v in ac.actor.state_dict().items()}
    cpu = {k: v.cpu() for k, v in data.items()}
    torch.set_num_threads(8)
    st = O.ppolag_update(ref, cpu, lam, [perm], batch_size=B, update_iters=1, kl_early_stop=False)
    torch.set_num_threads(1)
    up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False)
    out = up.run(data, torch.tensor([lam], device=DEV), perms=[perm], actor_lr=3e-4, critic_lr=3e-4)
    assert out['steps'] == M // B == len(st['loss_pi'])
    assert up.last_path == 'persistent'  # the kernel bench.py times, not the per-step fallback
    errs = {}
    for net in ('actor', 'reward_critic', 'cost_critic'):
        want = getattr(ref, net).state_dict()
        errs[net] = max(float((v.cpu() - want[k]).abs().max()) for k, v in getattr(ac, net).state_dict().items())
    moved = max(float((v.cpu() - init[k]).abs().max()) for k, v in ac.actor.state_dict().items())
    print('full-size pass vs oracle: max |theta - theta_oracle| =', errs, 'actor moved by', moved)
    assert moved > 0.01
    assert errs['actor'] <= 2e-6 and errs['reward_critic'] <= 2e-6 and errs['cost_critic'] <= 2e-6, errs
    s = out['stats'].double().cpu().numpy()
    l2 = 0.001
    for col, l2col, key in ((0, 5, 'loss_r'), (1, 6, 'loss_c'), (2, None, 'loss_pi')):
        mine = s[:, col] + (l2 * s[:, l2col] if l2col is not None else 0.0)
        want = np.asarray(st[key], np.float64)
        close = np.isclose(mine, want, rtol=2e-3, atol=2e-5)
        assert close.mean() >= 0.995, (key, close.mean())
        np.testing.assert_allclose(mine.mean(), want.mean(), rtol=1e-4, atol=1e-6, err_msg=key)
    np.testing.assert_allclose(s[:, 3], np.asarray(st['ratio_mean']), rtol=2e-4)
    np.testing.assert_allclose(out['kl'], st['kl'], rtol=2e-2)


@pytest.mark.parametrize('obs_dim,act_dim,kind', [(60, 2, 'focops'), (28, 8, 'focops'), (72, 2, 'p3o'), (44, 6, 'cup'),
                                                  (60, 17, 'focops')])
def test_extended_surrogates_pass_equals_per_minibatch_launches(obs_dim, act_dim, kind):
    """FOCOPS / CUP / P3O actor losses: the EXT instantiation of the persistent pass vs the per-step kernels
    (osa_ppo_minibatch_ext) on the same data, incl. wider action spaces (MFMA output tiles instead of the
    1-2-output VALU path) and a ragged last minibatch; the per-step kernels are pinned to the reference by the
    sibling goldens."""
    from omnisafe_amd.models import SurrogateExt
    from omnisafe_amd.update import PPOUpdater

    torch.manual_seed(obs_dim + act_dim)
    M, B = 1000, 64
    data = {'obs': torch.randn(M, obs_dim, device=DEV), 'act': torch.randn(M, act_dim, device=DEV),
            'target_value_r': torch.randn(M, device=DEV), 'target_value_c': torch.randn(M, device=DEV),
            'adv_r': torch.randn(M, device=DEV), 'adv_c': torch.randn(M, device=DEV)}
    perms = [torch.randperm(M), torch.randperm(M)]
    ext_kw = {'focops': dict(kl_coef=1.0, kl_mask_eta=2e-3, ratio_scale=1 / 1.5), 'cup': dict(kl_coef=1.0),
              'p3o': dict(cost_kappa=2.0, cost_excess=-0.05)}[kind]
    outs, acs = [], []
    for persistent in (True, False):
        torch.manual_seed(99)
        ac = make_ac(obs_dim, act_dim)
        if 'logp' not in data:
            _, _, _, lp = ac.step(data['obs'], eps=(data['act'] * 0))
            data['logp'] = lp + 0.2 * torch.randn(M, device=DEV)
        up = PPOUpdater(ac, batch_size=B, update_iters=2, target_kl=0.02, kl_early_stop=False, entropy_coef=0.01,
                        persistent=persistent, loss_kind=0 if kind == 'p3o' else 1, ext=SurrogateExt(**ext_kw),
                        update_critics=kind != 'cup')
        lam = torch.tensor([0.3], device=DEV)
        outs.append(up.run(data, lam, perms=perms, actor_lr=3e-3, critic_lr=1e-3))
        acs.append(ac)
    assert acs[0].adam_step.cpu().tolist() == acs[1].adam_step.cpu().tolist()
    for name in ('params', 'adam_m', 'adam_v'):
        a, b = getattr(acs[0], name).cpu().numpy(), getattr(acs[1], name).cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=2e-6, err_msg=name)
    s0, s1 = outs[0]['stats'].cpu().numpy(), outs[1]['stats'].cpu().numpy()
    np.testing.assert_allclose(s0[:, :5], s1[:, :5], rtol=2e-4, atol=2e-6)
    if kind == 'p3o':
        np.testing.assert_allclose(s0[:, 10], s1[:, 10], rtol=2e-4, atol=1e-6)
        assert (s0[:, 10] > 0).any()  # the penalty is active in some steps
    if kind == 'focops':  # the trust mask cuts in: with lr 3e-3 the per-sample KL crosses eta within the passes
        assert not np.allclose(s0[:, 2], 0)


@pytest.mark.parametrize('obs_dim,act_dim,B', [(376, 17, 64), (376, 17, 40), (200, 5, 128)])
def test_wide_input_minibatch_steps_vs_oracle(obs_dim, act_dim, B):
    """Wide observations (Humanoid 376 / 17: 24 input K blocks, beyond the persistent kernel) on the per-step
    kernels: three consecutive optimiser steps of all three networks vs the oracle (losses, parameters)."""
    from omnisafe_amd.update import PPOUpdater

    torch.manual_seed(obs_dim + B)
    M = 3 * B
    ref = O.ActorCritic(obs_dim, act_dim)
    ac = make_ac(obs_dim, act_dim)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        getattr(ac, net).load_state_dict(getattr(ref, net).state_dict())
    cpu = {'obs': torch.randn(M, obs_dim), 'act': torch.randn(M, act_dim), 'target_value_r': torch.randn(M),
           'target_value_c': torch.randn(M), 'adv_r': torch.randn(M), 'adv_c': torch.randn(M)}
    with torch.no_grad():
        cpu['logp'] = ref.actor.dist(cpu['obs']).log_prob(cpu['act']).sum(-1) + 0.2 * torch.randn(M)
    dev = {k: v.to(DEV) for k, v in cpu.items()}
    lam = 0.4
    up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False)
    up.hp.lr_actor = up.hp.lr_critic = 3e-4
    assert not up.lib.osa_ppo_pass_supported(obs_dim, act_dim, 64)
    perm = torch.randperm(M)
    for k in range(3):
        idx = perm[k * B:(k + 1) * B]
        l_r = O.critic_step(ref.reward_critic, ref.reward_critic_optimizer, cpu['obs'][idx], cpu['target_value_r'][idx])
        l_c = O.critic_step(ref.cost_critic, ref.cost_critic_optimizer, cpu['obs'][idx], cpu['target_value_c'][idx])
        l_p, ent, ratio = O.actor_step(ref.actor, ref.actor_optimizer, cpu['obs'][idx], cpu['act'][idx], cpu['logp'][idx],
                                       cpu['adv_r'][idx], cpu['adv_c'][idx], lam)
        stats = torch.zeros(16, device=DEV)
        up.minibatch(dev, idx.to(DEV), B, torch.tensor([lam], device=DEV), stats)
        s = stats.cpu().numpy()
        np.testing.assert_allclose(s[0] + 0.001 * s[5], l_r, rtol=2e-4)
        np.testing.assert_allclose(s[1] + 0.001 * s[6], l_c, rtol=2e-4)
        np.testing.assert_allclose(s[2], l_p, rtol=2e-3, atol=2e-5)
        np.testing.assert_allclose(s[3], float(ratio.mean()), rtol=2e-4)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in getattr(ac, net).state_dict().items():
            # (Adam's m / sqrt(v) turns summation-order noise on near-zero gradients into O(lr * 1e-2) steps:
            #  a handful of the 24 064 first-layer weights move by a few 1e-6 after three steps)
            np.testing.assert_allclose(v.cpu().numpy(), getattr(ref, net).state_dict()[k].numpy(), rtol=1e-4,
                                       atol=2e-5, err_msg=f'{net}/{k}')


@pytest.mark.parametrize('linear_lr_decay', [True, False])
@pytest.mark.parametrize('lr', [None, 1e-3])
def test_constraint_actor_critic_like_the_reference_test(linear_lr_decay, lr):
    """The reference's own test of this class (tests/test_model.py:127-178): a single unbatched observation
    through the module's call operator, output shapes, and the std annealing schedule
    (actor_critic.py:157-183: linear from 0.5 at epoch 1 to 0.1 at epoch 10, 0.1 outside)."""
    import types

    from omnisafe_amd.models import ConstraintActorCritic
    from omnisafe_amd.spaces import Box

    ns = types.SimpleNamespace
    obs_dim, act_dim = 10, 5
    model_cfgs = ns(weight_initialization_mode='kaiming_uniform', actor_type='gaussian_learning',
                    linear_lr_decay=linear_lr_decay, exploration_noise_anneal=False, std_range=[0.5, 0.1],
                    actor=ns(hidden_sizes=[64, 64], activation='tanh', lr=lr),
                    critic=ns(hidden_sizes=[64, 64], activation='tanh', lr=lr))
    cac = ConstraintActorCritic(obs_space=Box(low=-1.0, high=1.0, shape=(obs_dim,)),
                                act_space=Box(low=-1.0, high=1.0, shape=(act_dim,)), model_cfgs=model_cfgs,
                                epochs=10, device=DEV)
    obs = torch.randn(obs_dim, dtype=torch.float32)
    act, value_r, value_c, logp = cac(obs)
    assert act.shape == torch.Size([act_dim])
    assert value_r.shape == torch.Size([]) and value_c.shape == torch.Size([]) and logp.shape == torch.Size([])
    cac.set_annealing(epochs=[1, 10], std=[0.5, 0.1])
    cac.annealing(5)
    want = 0.5 + (5 - 1) / 9 * (0.1 - 0.5)
    assert cac.actor.std == pytest.approx(want, rel=1e-6)
    assert torch.allclose(cac.actor.log_std.cpu(), torch.full((act_dim,), float(np.log(want))), rtol=1e-6)
    for epoch, std in ((0, 0.1), (1, 0.5), (10, 0.1), (12, 0.1)):
        cac.annealing(epoch)
        assert cac.actor.std == pytest.approx(std, rel=1e-6)
    # the annealed std is what the next step samples with: logp of the mean action = -sum(log std) - d/2 log 2pi
    cac.annealing(5)
    act_det, _, _, logp_det = cac(obs, deterministic=True)
    assert float(logp_det) == pytest.approx(-act_dim * (np.log(want) + 0.5 * np.log(2 * np.pi)), rel=1e-5)


@pytest.mark.parametrize('obs_dim,act_dim,M,B,path', [(376, 17, 512, 64, 'persistent-wide-split'),
                                                      (60, 2, 1024, 128, 'persistent-chunked')])
def test_wrong_placement_is_caught_before_anything_changes(obs_dim, act_dim, M, B, path, monkeypatch):
    """The one-XCC variants of the cooperative passes rely on "workgroup b runs on XCC b mod 8".  Test hook
    OSA_DEBUG_PLACEMENT=wrong launches their protocol on the SPREAD grid: the kernels' placement check must trip
    before any parameter, moment or step counter is modified, the updater must repeat the pass with the spread
    variant, remember the verdict for the process, and the results must equal the per-step launches."""
    from omnisafe_amd import update as U
    from omnisafe_amd.update import PPOUpdater

    torch.manual_seed(3)
    data = {'obs': torch.randn(M, obs_dim, device=DEV), 'act': torch.randn(M, act_dim, device=DEV),
            'target_value_r': torch.randn(M, device=DEV), 'target_value_c': torch.randn(M, device=DEV),
            'adv_r': torch.randn(M, device=DEV), 'adv_c': torch.randn(M, device=DEV)}
    perms = [torch.randperm(M), torch.randperm(M)]
    keep = U._PLACEMENT['local_ok']
    acs, outs = [], []
    try:
        for persistent, debug in ((True, 'wrong'), (False, '')):
            U._PLACEMENT['local_ok'] = None
            monkeypatch.setenv('OSA_DEBUG_PLACEMENT', debug)
            monkeypatch.setenv('OSA_WIDE_SPLIT', 'local')
            monkeypatch.setenv('OSA_CHUNKED_PASS', '1')
            torch.manual_seed(99)
            ac = make_ac(obs_dim, act_dim)
            if 'logp' not in data:
                data['logp'] = ac.step(data['obs'], eps=(data['act'] * 0))[3] + 0.2 * torch.randn(M, device=DEV)
            up = PPOUpdater(ac, batch_size=B, update_iters=2, target_kl=0.02, kl_early_stop=False,
                            entropy_coef=0.01, persistent=persistent)
            outs.append(up.run(data, torch.tensor([0.3], device=DEV), perms=perms, actor_lr=3e-4, critic_lr=1e-3))
            acs.append(ac)
            if persistent:
                assert up.last_path == path
                assert U._PLACEMENT['local_ok'] is False  # the check tripped and the verdict was recorded
    finally:
        U._PLACEMENT['local_ok'] = keep
    assert acs[0].adam_step.cpu().tolist() == acs[1].adam_step.cpu().tolist()  # no step was counted twice
    for name in ('params', 'adam_m', 'adam_v'):
        a, b = getattr(acs[0], name).cpu().numpy(), getattr(acs[1], name).cpu().numpy()
        bad = np.abs(a - b) > 5e-6 + 1e-5 * np.abs(b)
        assert bad.sum() <= 4 and (not bad.any() or np.abs(a - b)[bad].max() <= 2.5e-4), name
    np.testing.assert_allclose(outs[0]['stats'].cpu().numpy()[:, :10], outs[1]['stats'].cpu().numpy()[:, :10],
                               rtol=1e-5, atol=2e-7)
