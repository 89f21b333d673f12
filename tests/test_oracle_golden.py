"""CPU tests: the oracle (oracle/np_oracle.py) against golden vectors produced by the unmodified
reference (oracle/make_golden.py) and against the reference's own known-answer tests."""
import numpy as np
import pytest
import torch

import np_oracle as O


# ---- reference's own KATs: tests/test_utils.py:95-115 (hard-coded fp64 answers) ---------------
KAT = {
    0.9: [11.4265, 11.585, 10.65, 8.5, 5.0],
    0.99: [14.604476, 13.741895, 11.8605, 8.95, 5.0],
    0.999: [14.96004, 13.974015, 11.985985, 8.995, 5.0],
}


@pytest.mark.parametrize('g', [0.9, 0.99, 0.999])
def test_discount_cumsum_reference_kat(g):
    y = O.discount_cumsum(np.array([1, 2, 3, 4, 5], np.float32), g)
    assert np.allclose(y, KAT[g], rtol=1e-5, atol=1e-8)


def test_discount_cumsum_golden(golden):
    gd = golden('discount_cumsum.npz')
    for g in (0.9, 0.99, 0.999):
        y = O.discount_cumsum(np.array([1, 2, 3, 4, 5], np.float32), g)
        assert np.array_equal(y, gd[f'kat_{g}'])
    y = O.discount_cumsum(gd['x_rand'], 0.99 * 0.95)
    assert np.array_equal(y, gd['y_rand_0.9405'])  # bit-exact fp64


# ---- CPO case ids: reference tests/test_policy.py:55-74 ----------------------------------------
@pytest.mark.parametrize('b,c,q,r,s,case', [
    ([1., 1.], -1., 1., 1., 1., 3),       # ep_costs<0, B<0
    ([1., 1.], -0.01, 1., 1., 1., 2),     # ep_costs<0, B>=0
    ([1., 1.], 0.01, 1., 1., 1., 1),      # ep_costs>=0, B>=0
    ([1., 1.], 1., 1., 1., 1., 0),        # ep_costs>=0, B<0
    ([1e-4, 1e-4], -1., 1., 1., 1., 4),   # tiny cost gradient
])
def test_cpo_case_ids(b, c, q, r, s, case):
    oc, A, B = O.cpo_determine_case(torch.tensor(b), torch.tensor(c), torch.tensor(q),
                                    torch.tensor(r), torch.tensor(s), target_kl=0.01)
    assert oc == case


# ---- buffer / GAE ----------------------------------------------------------------------------
@pytest.mark.parametrize('est', ['gae', 'gae-rtg', 'plain', 'vtrace'])
@pytest.mark.parametrize('pc', [0.0, 0.3])
def test_gae_bit_exact_vs_reference(golden, est, pc):
    g = golden('buffer.npz')
    args = (g['reward'], g['cost'], g['value_r'], g['value_c'], g['path_end'], g['boot_r'],
            g['boot_c'], float(g['gamma']), float(g['lam']), float(g['lam_c']), pc, est)
    tm = O.gae_time_major(*args)
    pp = O.gae_per_path(*args)
    tag = f'{est}_pc{pc}'
    for ok, gk in (('adv_r', 'adv_r'), ('adv_c', 'adv_c'), ('tgt_r', 'target_value_r'),
                   ('tgt_c', 'target_value_c'), ('disc_ret', 'discounted_ret')):
        ref = g[f'{tag}/raw/{gk}']
        assert np.array_equal(O.env_major(pp[ok]), ref), (ok, 'per-path')
        assert np.array_equal(O.env_major(tm[ok]), ref), (ok, 'time-major')


def test_buffer_get_vs_reference(golden):
    g = golden('buffer.npz')
    tm = O.gae_time_major(g['reward'], g['cost'], g['value_r'], g['value_c'], g['path_end'],
                          g['boot_r'], g['boot_c'], 0.99, 0.95, 0.9)
    a_r, a_c, _ = O.buffer_get(tm['adv_r'], tm['adv_c'])
    assert np.array_equal(a_r, g['gae_pc0.0/get/adv_r'])
    assert np.array_equal(a_c, g['gae_pc0.0/get/adv_c'])
    for k in ('obs', 'act', 'logp'):
        assert np.array_equal(O.env_major(g[k]), g[f'gae_pc0.0/get/{k}'])


# ---- normaliser ------------------------------------------------------------------------------
def test_normalizer_vs_reference(golden):
    g = golden('normalizer.npz')
    norm = O.Normalizer((7,), clip=5)
    for i in range(int(g['n_batches'])):
        y = norm.normalize(torch.from_numpy(g[f'in{i}'].copy()))
        assert np.array_equal(y.numpy(), g[f'out{i}']), i
        assert np.array_equal(norm.mean.numpy(), g[f'mean{i}'])
        assert np.array_equal(norm.sumsq.numpy(), g[f'sumsq{i}'])
        assert np.array_equal(norm.std.numpy(), g[f'std{i}'], equal_nan=True)  # count==1 -> 0/0
        assert norm.count == int(g[f'count{i}'])


# ---- actor-critic step -----------------------------------------------------------------------
def load_ac(g, prefix, obs_dim=60, act_dim=2, **kw):
    ac = O.ActorCritic(obs_dim, act_dim, **kw)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        sd = {k[len(prefix) + len(net) + 1:]: torch.from_numpy(v.copy()) for k, v in g.items()
              if k.startswith(f'{prefix}{net}/')}
        getattr(ac, net).load_state_dict(sd)
    return ac


def test_actor_critic_step_vs_reference(golden):
    g = golden('actor_critic_step.npz')
    ac = load_ac(g, '')
    act, v_r, v_c, logp = ac.step(torch.from_numpy(g['obs']), eps=torch.from_numpy(g['eps']))
    assert np.array_equal(act.numpy(), g['act'])
    assert np.array_equal(v_r.numpy(), g['value_r'])
    assert np.array_equal(v_c.numpy(), g['value_c'])
    assert np.array_equal(logp.numpy(), g['logp'])
    a_det, _, _, lp_det = ac.step(torch.from_numpy(g['obs']), deterministic=True)
    assert np.array_equal(a_det.numpy(), g['act_det'])
    assert np.array_equal(lp_det.numpy(), g['logp_det'])
    # parameter order is log_std first (checkpoint key order of the reference)
    assert [k for k, _ in ac.actor.named_parameters()][0] == 'log_std'


# ---- whole epoch: rollout on the recorded trace, then PPOLag update ---------------------------
def _trace(g):
    return {'reset_obs': g['rollout/reset_obs'], 'obs': g['rollout/obs'], 'reward': g['rollout/reward'],
            'cost': g['rollout/cost'], 'terminated': g['rollout/terminated'],
            'truncated': g['rollout/truncated'], 'final_obs': g['rollout/final_obs'],
            'eps': g['rollout/eps']}


def test_rollout_vs_reference(golden):
    g = golden('ppolag_epoch.npz')
    torch.set_num_threads(1)
    ac = load_ac(g, 'init/')
    norm = O.Normalizer((60,), clip=5)
    buf, gae, aux = O.rollout_on_trace(ac, norm, _trace(g))
    for k in ('obs', 'act', 'reward', 'cost', 'value_r', 'value_c', 'logp'):
        assert np.array_equal(buf[k], g[f'buffer/{k}']), k
    for ok, gk in (('adv_r', 'adv_r'), ('adv_c', 'adv_c'), ('tgt_r', 'target_value_r'),
                   ('tgt_c', 'target_value_c'), ('disc_ret', 'discounted_ret')):
        assert np.array_equal(gae[ok], g[f'buffer/{gk}']), ok
    assert np.array_equal(norm.mean.numpy(), g['rollout/norm_mean'])
    assert np.array_equal(norm.std.numpy(), g['rollout/norm_std'])
    assert norm.count == int(g['rollout/norm_count'])
    assert np.array_equal(aux['episodes'][:, 0], g['rollout/ep_ret_window'])
    assert np.array_equal(aux['episodes'][:, 1], g['rollout/ep_cost_window'])
    assert np.array_equal(aux['episodes'][:, 2], g['rollout/ep_len_window'])


def _update_data(g):
    a_r, a_c, _ = O.buffer_get(g['buffer/adv_r'], g['buffer/adv_c'])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    return {'obs': t(O.env_major(g['buffer/obs'])), 'act': t(O.env_major(g['buffer/act'])),
            'logp': t(O.env_major(g['buffer/logp'])),
            'target_value_r': t(O.env_major(g['buffer/target_value_r'])),
            'target_value_c': t(O.env_major(g['buffer/target_value_c'])),
            'adv_r': t(a_r), 'adv_c': t(a_c)}


def test_ppolag_update_vs_reference(golden):
    g = golden('ppolag_epoch.npz')
    torch.set_num_threads(1)
    ac = load_ac(g, 'init/')
    lag = O.Lagrange(cost_limit=25.0, lagrangian_multiplier_init=0.001, lambda_lr=0.035)
    assert np.float32(lag.lagrangian_multiplier.item()) == g['update/lambda_before']
    lag.update_lagrange_multiplier(float(g['update/Jc']))
    assert np.float32(lag.lagrangian_multiplier.item()) == g['update/lambda_after']
    stats = O.ppolag_update(ac, _update_data(g), lag.lagrangian_multiplier.item(), g['update/perms'],
                            batch_size=64, update_iters=3, kl_early_stop=False)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in getattr(ac, net).state_dict().items():
            assert np.array_equal(v.numpy(), g[f'post/{net}/{k}']), (net, k)
    assert np.allclose(stats['loss_pi'], g['update/loss_pi'], rtol=0, atol=0)
    assert np.allclose(stats['loss_r'], g['update/loss_r'], rtol=0, atol=0)
    assert np.allclose(stats['loss_c'], g['update/loss_c'], rtol=0, atol=0)
    assert np.float32(stats['kl']) == g['update/kl'][-1]
    assert stats['stop_iter'] == int(g['update/stop_iter'][-1])


def test_reach_env_statement_known_answers():
    """Hand-computed transitions of the learnable stand-in task (oracle/np_oracle.py:reach_env_step)."""
    f = np.float32
    state = np.array([
        [0.0, 0.0, 0.3, 0.4, 1.0, 1.0],     # moves towards the goal, far from the hazard
        [0.0, 0.0, 0.1, 0.0, 0.05, 0.0],    # reaches the goal and sits in the hazard
        [1.45, -1.45, 0.0, 0.0, -1.0, 1.0],  # clipped at the wall, action clipped to [-1, 1]
    ], dtype=f)
    action = np.array([[0.6, 0.8], [1.0, 0.0], [3.0, -3.0]], dtype=f)
    q, r, c, reached = O.reach_env_step(state, action)
    np.testing.assert_allclose(q, [[0.06, 0.08], [0.1, 0.0], [1.5, -1.5]], rtol=0, atol=1e-7)
    np.testing.assert_allclose(r[0], 0.5 - 0.4, atol=1e-6)          # |(.3,.4)| = .5 -> |(.24,.32)| = .4
    assert reached.tolist() == [False, True, False]
    np.testing.assert_allclose(r[1], 0.1 + 1.0, atol=1e-6)           # progress 0.1 plus the goal bonus
    assert c.tolist() == [0.0, 1.0, 0.0]
    assert r[2] < 0                                                  # pushed away from the goal at the origin
    obs = O.reach_env_obs(np.array([[0.1, 0.2, 0.5, 0.1, -0.3, 0.0]], f), 60)
    np.testing.assert_allclose(obs[0, :6], [0.1, 0.2, 0.4, -0.1, -0.4, -0.2], atol=1e-7)
    assert obs.shape == (1, 60) and not obs[0, 6:].any()


def test_learning_golden_file_is_complete():
    import json
    import os

    g = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'learning_reach.json')))
    cfg = g['config']
    assert cfg['env_id'] == 'SynthReach-v0' and cfg['epochs'] == 10
    for algo in ('PPOLag', 'TRPOLag', 'CPO'):
        curves = g['curves'][algo]
        # 20 seeds per algorithm; CPO has 80 (round 2: large-sample follow-up of its episode-cost tail)
        assert sorted(map(int, curves)) == list(range(80 if algo == 'CPO' else 20))
        for c in curves.values():
            assert len(c['EpRet']) == len(c['EpCost']) == cfg['epochs']
            assert np.isfinite(c['EpRet']).all() and np.isfinite(c['EpCost']).all()
    # the reference learns the task: PPOLag's seed-mean return rises monotonically from ~0 to ~7
    m = np.mean([c['EpRet'] for c in g['curves']['PPOLag'].values()], axis=0)
    assert abs(m[0]) < 0.2 and m[-1] > 6 and np.all(np.diff(m) > 0)


# ------------------------------------------------------------------ data parallelism: the 2-rank reference goldens
def _dp2_datas(g, world):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    return [{k: t(g[f'r{r}/data/{k}']) for k in ('obs', 'act', 'logp', 'target_value_r', 'target_value_c', 'adv_r',
                                                  'adv_c')} for r in range(world)]


@pytest.mark.parametrize('tag,obs_dim,act_dim,bs', [('dp2_ppolag_point', 60, 2, 64),
                                                    ('dp2_ppolag_humanoid', 376, 17, 64),
                                                    ('dp2_ppolag_point_largebatch', 60, 2, 2048)])
def test_dp2_ppolag_update_vs_reference(golden, tag, obs_dim, act_dim, bs):
    """tests/golden/dp2_*.npz = one `_update()` of the UNMODIFIED reference on TWO ranks (gloo; oracle/make_golden.py::
    gen_dp2_updates).  The oracle's data-parallel restatement -- local clip, then (g_0 + g_1) / 2 per parameter, one
    Adam step; KL averaged over the ranks (policy_gradient.py:390, 437-442; utils/distributed.py:167-198) -- must
    reproduce the reference's post-update parameters BIT FOR BIT (a two-operand float sum is commutative)."""
    g = golden(f'{tag}.npz')
    world = int(g['world'])
    assert world == 2
    torch.set_num_threads(1)
    ac = load_ac(g, 'init/', obs_dim, act_dim)
    lag = O.Lagrange(cost_limit=0.5, lagrangian_multiplier_init=0.5, lambda_lr=0.035)
    lag.update_lagrange_multiplier(float(g['Jc']))
    assert np.float32(lag.lagrangian_multiplier.item()) == g['lambda_after']
    # Jc is the cross-rank mean of the per-rank window means (logger.py:359-374 through dist_statistics_scalar)
    jc = np.mean([g[f'r{r}/ep_cost_window'].mean(dtype=np.float64) for r in range(world)])
    np.testing.assert_allclose(jc, float(g['Jc']), rtol=1e-6)
    perms = [g[f'r{r}/perms'] for r in range(world)]
    assert not np.array_equal(perms[0], perms[1])  # per-rank seeds -> per-rank minibatch orders
    stats = O.ppolag_update_dp(ac, _dp2_datas(g, world), lag.lagrangian_multiplier.item(), perms, batch_size=bs,
                               update_iters=2, kl_early_stop=False)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in getattr(ac, net).state_dict().items():
            assert np.array_equal(v.numpy(), g[f'post/{net}/{k}']), (net, k, float(np.abs(v.numpy() - g[f'post/{net}/{k}']).max()))
    for r in range(world):  # every rank logs ITS losses
        np.testing.assert_array_equal(np.float32([s[r] for s in stats['loss_pi']]), g[f'r{r}/log/Loss/Loss_pi'])
        np.testing.assert_array_equal(np.float32([s[r] for s in stats['loss_r']]),
                                      g[f'r{r}/log/Loss/Loss_reward_critic'])
    assert np.float32(stats['kl']) == g['r0/log/Train/KL'][-1] == g['r1/log/Train/KL'][-1]


def test_dp2_trpolag_update_vs_reference(golden):
    """BASELINE config 5's algorithm under two ranks: one `_update()` of the UNMODIFIED reference (TRPOLag, SynthAnt 27 / 8,
    gloo) against the oracle's data-parallel restatement -- policy gradient and every Fisher-vector product averaged over
    the ranks, the line search on the rank averages, then batch-128 critic steps clipped per rank and averaged
    (natural_pg.py:91-119, 185-240; base/trpo.py:93-222; utils/distributed.py:167-198).  Everything is float32 tensor
    arithmetic in the reference's order (a two-operand float sum is commutative): the post-update parameters of all
    three networks, the trust-region step's logged scalars and every rank's critic losses BIT FOR BIT."""
    g = golden('dp2_trpolag_ant.npz')
    world = int(g['world'])
    assert world == 2 and str(g['algo']) == 'TRPOLag'
    torch.set_num_threads(1)
    ac = load_ac(g, 'init/', 27, 8, actor_lr=None, critic_lr=1e-3)  # (TRPOLag.yaml: no actor optimiser, critic lr 0.001)
    lag = O.Lagrange(cost_limit=0.5, lagrangian_multiplier_init=0.5, lambda_lr=0.035)
    lag.update_lagrange_multiplier(float(g['Jc']))
    assert np.float32(lag.lagrangian_multiplier.item()) == g['lambda_after']
    perms = [g[f'r{r}/perms'] for r in range(world)]
    stats = O.trpolag_update_dp(ac, _dp2_datas(g, world), lag.lagrangian_multiplier.item(), perms, batch_size=128,
                                update_iters=perms[0].shape[0])
    assert stats['acceptance_step'] == int(g['r0/log/Misc/AcceptanceStep'][0]) == int(g['r1/log/Misc/AcceptanceStep'][0])
    for key, name in (('xHx', 'Misc/xHx'), ('alpha', 'Misc/Alpha'), ('final_step_norm', 'Misc/FinalStepNorm'),
                      ('gradient_norm', 'Misc/gradient_norm')):
        assert np.float32(stats[key]) == g[f'r0/log/{name}'][0] == g[f'r1/log/{name}'][0], key
    assert np.float32(stats['kl']) == g['r0/log/Train/KL'][-1] == g['r1/log/Train/KL'][-1]
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in getattr(ac, net).state_dict().items():
            assert np.array_equal(v.numpy(), g[f'post/{net}/{k}']), (net, k, float(np.abs(v.numpy() - g[f'post/{net}/{k}']).max()))
    for r in range(world):
        np.testing.assert_array_equal(np.float32([s[r] for s in stats['loss_r']]), g[f'r{r}/log/Loss/Loss_reward_critic'])
        np.testing.assert_array_equal(np.float32([s[r] for s in stats['loss_c']]), g[f'r{r}/log/Loss/Loss_cost_critic'])


def test_dp2_cpo_update_vs_reference(golden):
    """BASELINE config 3's algorithm under two ranks: one `_update()` of the UNMODIFIED reference (CPO, SynthCarGoal 72 / 2,
    cost limit below the episode cost: the infeasible-recovery case) against `cpo_update_dp` -- both policy gradients and
    every Fisher-vector product of both conjugate-gradient solves averaged over the ranks, the case analysis and the
    two-constraint line search on rank averages (second_order/cpo.py:57-462), clip-then-average critic steps: the
    optimisation case, every logged scalar of the step and the post-update parameters of all three networks BIT FOR
    BIT."""
    g = golden('dp2_cpo_car.npz')
    world = int(g['world'])
    assert world == 2 and str(g['algo']) == 'CPO'
    torch.set_num_threads(1)
    ac = load_ac(g, 'init/', 72, 2, actor_lr=None, critic_lr=1e-3)
    perms = [g[f'r{r}/perms'] for r in range(world)]
    ep_costs = float(g['Jc']) - 0.5
    stats = O.cpo_update_dp(ac, _dp2_datas(g, world), ep_costs, perms, batch_size=128, update_iters=perms[0].shape[0])
    assert stats['optim_case'] == int(g['r0/log/Misc/OptimCase'][0]) == 0  # c^2 / s - 2 delta > 0 and c > 0: recovery
    assert stats['acceptance_step'] == int(g['r0/log/Misc/AcceptanceStep'][0])
    for key, name in (('xHx', 'xHx'), ('alpha', 'Alpha'), ('q', 'q'), ('r', 'r'), ('s', 's'), ('A', 'A'), ('B', 'B'),
                      ('lambda_star', 'Lambda_star'), ('nu_star', 'Nu_star'), ('final_step_norm', 'FinalStepNorm'),
                      ('gradient_norm', 'gradient_norm'), ('cost_gradient_norm', 'cost_gradient_norm')):
        assert np.float32(stats[key]) == g[f'r0/log/Misc/{name}'][0] == g[f'r1/log/Misc/{name}'][0], key
    assert np.float32(stats['kl']) == g['r0/log/Train/KL'][-1]
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in getattr(ac, net).state_dict().items():
            assert np.array_equal(v.numpy(), g[f'post/{net}/{k}']), (net, k, float(np.abs(v.numpy() - g[f'post/{net}/{k}']).max()))
    for r in range(world):
        np.testing.assert_array_equal(np.float32([s[r] for s in stats['loss_r']]), g[f'r{r}/log/Loss/Loss_reward_critic'])
        np.testing.assert_array_equal(np.float32([s[r] for s in stats['loss_c']]), g[f'r{r}/log/Loss/Loss_cost_critic'])


def test_dp4_updates_vs_four_rank_reference(golden):
    """FOUR (and, for PPOLag, EIGHT) ranks of the unmodified reference (`oracle/make_golden.py dp4` / `dp8`): a sum over four ranks is no longer order-free
    (gloo's ring adds in its own order, the restatement in rank order), so the post-update parameters agree to float32
    round-off (measured 3e-8) instead of bit for bit; the line-search decision and the multiplier are identical."""
    torch.set_num_threads(1)
    g = golden('dp4_ppolag_point.npz')
    assert int(g['world']) == 4
    ac = load_ac(g, 'init/', 60, 2)
    lag = O.Lagrange(cost_limit=0.5, lagrangian_multiplier_init=0.5, lambda_lr=0.035)
    lag.update_lagrange_multiplier(float(g['Jc']))
    assert np.float32(lag.lagrangian_multiplier.item()) == g['lambda_after']
    O.ppolag_update_dp(ac, _dp2_datas(g, 4), lag.lagrangian_multiplier.item(), [g[f'r{r}/perms'] for r in range(4)],
                       batch_size=64, update_iters=2, kl_early_stop=False)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in getattr(ac, net).state_dict().items():
            np.testing.assert_allclose(v.numpy(), g[f'post/{net}/{k}'], rtol=0, atol=2e-7, err_msg=f'{net}/{k}')
    g = golden('dp8_ppolag_point.npz')  # (EIGHT ranks: the world size BASELINE.json quotes)
    assert int(g['world']) == 8
    ac = load_ac(g, 'init/', 60, 2)
    lag = O.Lagrange(cost_limit=0.5, lagrangian_multiplier_init=0.5, lambda_lr=0.035)
    lag.update_lagrange_multiplier(float(g['Jc']))
    assert np.float32(lag.lagrangian_multiplier.item()) == g['lambda_after']
    O.ppolag_update_dp(ac, _dp2_datas(g, 8), lag.lagrangian_multiplier.item(), [g[f'r{r}/perms'] for r in range(8)],
                       batch_size=64, update_iters=2, kl_early_stop=False)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in getattr(ac, net).state_dict().items():
            np.testing.assert_allclose(v.numpy(), g[f'post/{net}/{k}'], rtol=0, atol=2e-7, err_msg=f'{net}/{k}')
    g = golden('dp4_trpolag_ant.npz')
    ac = load_ac(g, 'init/', 27, 8, actor_lr=None, critic_lr=1e-3)
    lag = O.Lagrange(cost_limit=0.5, lagrangian_multiplier_init=0.5, lambda_lr=0.035)
    lag.update_lagrange_multiplier(float(g['Jc']))
    stats = O.trpolag_update_dp(ac, _dp2_datas(g, 4), lag.lagrangian_multiplier.item(),
                                [g[f'r{r}/perms'] for r in range(4)], batch_size=128, update_iters=2)
    assert stats['acceptance_step'] == int(g['r0/log/Misc/AcceptanceStep'][0])
    np.testing.assert_allclose(stats['xHx'], g['r0/log/Misc/xHx'][0], rtol=1e-5)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in getattr(ac, net).state_dict().items():
            np.testing.assert_allclose(v.numpy(), g[f'post/{net}/{k}'], rtol=0, atol=2e-7, err_msg=f'{net}/{k}')


def test_dp8_config_shapes_vs_eight_rank_reference(golden):
    """BASELINE configs 4, 5 and 3 at the world size BASELINE.json quotes them on: EIGHT ranks of the unmodified reference
    (`oracle/make_golden.py dp8 dp2_ppolag_humanoid@32 dp2_trpolag_ant dp2_cpo_car`) against the oracle's data-parallel
    restatements -- rank-ordered eight-term sums v gloo's ring order: float32 round-off instead of bit for bit; the
    accepted line-search index, CPO's case and the multiplier identical."""
    torch.set_num_threads(1)
    g = golden('dp8_ppolag_humanoid.npz')
    assert int(g['world']) == 8 and int(g['T']) == 32
    ac = load_ac(g, 'init/', 376, 17)
    lag = O.Lagrange(cost_limit=0.5, lagrangian_multiplier_init=0.5, lambda_lr=0.035)
    lag.update_lagrange_multiplier(float(g['Jc']))
    assert np.float32(lag.lagrangian_multiplier.item()) == g['lambda_after']
    O.ppolag_update_dp(ac, _dp2_datas(g, 8), lag.lagrangian_multiplier.item(), [g[f'r{r}/perms'] for r in range(8)],
                       batch_size=64, update_iters=2, kl_early_stop=False)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in getattr(ac, net).state_dict().items():
            np.testing.assert_allclose(v.numpy(), g[f'post/{net}/{k}'], rtol=0, atol=5e-7, err_msg=f'{net}/{k}')
    g = golden('dp8_trpolag_ant.npz')
    ac = load_ac(g, 'init/', 27, 8, actor_lr=None, critic_lr=1e-3)
    lag = O.Lagrange(cost_limit=0.5, lagrangian_multiplier_init=0.5, lambda_lr=0.035)
    lag.update_lagrange_multiplier(float(g['Jc']))
    assert np.float32(lag.lagrangian_multiplier.item()) == g['lambda_after']
    stats = O.trpolag_update_dp(ac, _dp2_datas(g, 8), lag.lagrangian_multiplier.item(),
                                [g[f'r{r}/perms'] for r in range(8)], batch_size=128, update_iters=2)
    assert stats['acceptance_step'] == int(g['r0/log/Misc/AcceptanceStep'][0])
    np.testing.assert_allclose(stats['xHx'], g['r0/log/Misc/xHx'][0], rtol=1e-4)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in getattr(ac, net).state_dict().items():
            np.testing.assert_allclose(v.numpy(), g[f'post/{net}/{k}'], rtol=0, atol=5e-7, err_msg=f'{net}/{k}')
    g = golden('dp8_cpo_car.npz')
    ac = load_ac(g, 'init/', 72, 2, actor_lr=None, critic_lr=1e-3)
    perms = [g[f'r{r}/perms'] for r in range(8)]
    stats = O.cpo_update_dp(ac, _dp2_datas(g, 8), float(g['Jc']) - 0.5, perms, batch_size=128,
                            update_iters=perms[0].shape[0])
    assert stats['optim_case'] == int(g['r0/log/Misc/OptimCase'][0])
    assert stats['acceptance_step'] == int(g['r0/log/Misc/AcceptanceStep'][0])
    for key, name in (('xHx', 'xHx'), ('q', 'q'), ('r', 'r'), ('s', 's'), ('nu_star', 'Nu_star')):
        np.testing.assert_allclose(stats[key], g[f'r0/log/Misc/{name}'][0], rtol=1e-3, err_msg=key)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in getattr(ac, net).state_dict().items():
            np.testing.assert_allclose(v.numpy(), g[f'post/{net}/{k}'], rtol=0, atol=1e-6, err_msg=f'{net}/{k}')


@pytest.mark.parametrize('tag', ['dp2_ppolag_point', 'dp2_trpolag_ant', 'dp2_cpo_car', 'dp4_ppolag_point',
                                 'dp8_ppolag_point', 'dp8_ppolag_humanoid', 'dp8_trpolag_ant', 'dp8_cpo_car'])
def test_dp2_advantage_statistics_vs_reference(golden, tag):
    """VectorOnPolicyBuffer.get() on two (four, eight) ranks: the advantages every rank hands to `_update()` are standardised with
    the GLOBAL mean / population std (utils/distributed.py:382-392)."""
    g = golden(f'{tag}.npz')
    world = int(g['world'])
    raw_r = [torch.from_numpy(O.env_major(g[f'r{r}/raw/adv_r']).copy()) for r in range(world)]
    raw_c = [torch.from_numpy(O.env_major(g[f'r{r}/raw/adv_c']).copy()) for r in range(world)]
    a_r, a_c, (mr, sr, mc) = O.dp_standardise(raw_r, raw_c)
    for r in range(world):
        np.testing.assert_allclose(a_r[r].numpy(), g[f'r{r}/data/adv_r'], rtol=0, atol=2e-6)
        np.testing.assert_allclose(a_c[r].numpy(), g[f'r{r}/data/adv_c'], rtol=0, atol=2e-6)
    # ... and NOT with per-rank statistics (the comparison is not vacuous)
    local = (raw_r[0] - raw_r[0].mean()) / (raw_r[0].std(unbiased=False) + 1e-8)
    assert float((local - torch.from_numpy(g['r0/data/adv_r'])).abs().max()) > 1e-3
