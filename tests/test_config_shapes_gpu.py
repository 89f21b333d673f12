"""GPU parity at the observation / action shapes of EVERY BASELINE.json config, M = 4096 transitions, against
one whole `_update()` of the unmodified reference per config (tests/golden/config{2..5}_*.npz, written by
oracle/make_golden.py::gen_config_shape_updates): same initial parameters, same `buf.get()` output, same
EpCost window, same minibatch permutations -> parameters of all three networks, multiplier, logged
statistics.

  config 2  PPOLag  60/2    128 chained Adam steps of 64 rows          (persistent pass kernel)
  config 3  CPO     72/2    CG + FVP + CPO case algebra + line search, then 64 critic steps of 128 rows
  config 4  PPOLag  376/17  the wide-input path (W1 does not fit the 64x96 LDS tile of the narrow kernel)
  config 5  TRPOLag 27/8    rows that are not 16-byte aligned (padded once per update), D_a = 8

Tolerances (float32; the reference sums in CPU-sgemm order, the kernels in MFMA-tile order).  Measured on
MI355X (round 2): max |theta - theta_reference| = 4e-8 (config 2), 1.6e-7 (config 4), 3e-8 / 4e-6 (config 5
actor / critics), <1e-7 (config 3).  Required, with ~10x margin:
  first-order family: parameters after 128 chained Adam steps  atol 2e-6; losses rtol 2e-3; KL rtol 1e-2;
  trust-region family: accepted line-search index and CPO case identical; actor atol 5e-5 (theta_old + a step
    of norm ~0.33 whose direction comes out of 15 float32 CG iterations); critics atol 2e-5.
"""
import numpy as np
import pytest
import torch

from test_siblings_gpu import _check_loss_pi_call_log, _check_params, _log, _run_update

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

LAG = {'lagrangian_multiplier_init': 0.5, 'cost_limit': 0.5}
FIRST_ORDER = [('config2_ppolag_point', 'PPOLag', 'SynthPointGoal1-v0', ({}, LAG)),
               ('config4_ppolag_humanoid', 'PPOLag', 'SynthHumanoid-v0', ({}, LAG))]
TRUST_REGION = [('config3_cpo_car', 'CPO', 'SynthCarGoal1-v0', ({'cost_limit': 0.71}, None)),
                ('config3_cpo_car_case0', 'CPO', 'SynthCarGoal1-v0', ({'cost_limit': 0.5}, None)),
                ('config5_trpolag_ant', 'TRPOLag', 'SynthAnt-v0', ({}, LAG))]


def _max_err(ac, g, net):
    return max(float(np.abs(v.cpu().numpy() - g[f'post/{net}/{k}']).max())
               for k, v in getattr(ac, net).state_dict().items())


@pytest.mark.parametrize('tag,name,env_id,extra', FIRST_ORDER)
def test_first_order_update_at_config_shape(golden, tmp_path, tag, name, env_id, extra):
    g = golden(f'{tag}.npz')
    assert g['data/obs'].shape[0] == 4096
    algo, ac = _run_update(name, tag, g, tmp_path, trust_region=False, env_id=env_id, extra=extra)
    assert algo._last_update_steps == 2 * 64
    print(tag, 'max |param - reference|:', {n: _max_err(ac, g, n) for n in ('actor', 'reward_critic', 'cost_critic')})
    _check_params(ac, g, ('actor', 'reward_critic', 'cost_critic'), 2e-6)
    # the parameters moved by much more than the tolerance (the comparison is not vacuous)
    moved = max(float(np.abs(g[f'post/actor/{k}'] - g[f'init/actor/{k}']).max()) for k in ac.actor.state_dict())
    assert moved > 5e-3
    np.testing.assert_allclose(algo._lagrange.lagrangian_multiplier, float(g['lambda_after']), rtol=1e-6)
    np.testing.assert_allclose(_log(algo, 'Train/KL')[-1], g['log/Train/KL'][-1], rtol=1e-2, atol=1e-7)
    np.testing.assert_allclose(_log(algo, 'Loss/Loss_pi').mean(), g['log/Loss/Loss_pi'].mean(), rtol=2e-3,
                               atol=2e-6)
    for key in ('Loss/Loss_reward_critic', 'Loss/Loss_cost_critic'):
        np.testing.assert_allclose(_log(algo, key).mean(), g['log/' + key].mean(), rtol=2e-4)
    np.testing.assert_allclose(_log(algo, 'Train/Entropy').mean(), g['log/Train/Entropy'].mean(), rtol=1e-5)
    # Value/Adv: the reference's loop variable shadows the full batch, so it logs the LAST minibatch's mean advantage
    np.testing.assert_allclose(_log(algo, 'Value/Adv')[-1], g['log/Value/Adv'][-1], rtol=1e-5, atol=1e-7)
    assert abs(float(g['log/Value/Adv'][-1]) - float(g['data/adv_r'].mean())) > 1e-3  # (not the full-batch mean)


@pytest.mark.parametrize('tag,name,env_id,extra', TRUST_REGION)
def test_trust_region_update_at_config_shape(golden, tmp_path, tag, name, env_id, extra):
    g = golden(f'{tag}.npz')
    assert g['data/obs'].shape[0] == 4096
    algo, ac = _run_update(name, tag, g, tmp_path, trust_region=True, env_id=env_id, extra=extra)
    print(tag, 'max |param - reference|:', {n: _max_err(ac, g, n) for n in ('actor', 'reward_critic', 'cost_critic')})
    assert int(_log(algo, 'Misc/AcceptanceStep')[-1]) == int(g['log/Misc/AcceptanceStep'][-1])
    for key, rtol in (('Misc/Alpha', 1e-2), ('Misc/xHx', 1e-2), ('Misc/gradient_norm', 1e-3),
                      ('Misc/H_inv_g', 1e-2), ('Misc/FinalStepNorm', 2e-2)):
        np.testing.assert_allclose(_log(algo, key)[-1], g['log/' + key][-1], rtol=rtol, err_msg=key)
    if name == 'CPO':
        info = algo._last_actor_update
        want_case = 0 if tag.endswith('case0') else 1
        assert info['case'] == int(g['log/Misc/OptimCase'][-1]) == want_case
        # case 1: constraint violated but recoverable, both projections (lambda_a, lambda_b) are evaluated and
        # nu* > 0 (cpo.py:300-322); case 0: infeasible, pure recovery step along -F^-1 b (cpo.py:289-299, 377-383)
        assert float(g['log/Misc/Nu_star'][-1]) > 1.0
        np.testing.assert_allclose(_log(algo, 'Misc/Nu_star')[-1], g['log/Misc/Nu_star'][-1], rtol=5e-2)
        for key, rtol in (('Misc/q', 1e-2), ('Misc/r', 5e-2), ('Misc/s', 1e-2), ('Misc/cost_gradient_norm', 1e-3),
                          ('Misc/A', 1e-2), ('Misc/B', 5e-2), ('Misc/Lambda_star', 5e-2)):
            np.testing.assert_allclose(_log(algo, key)[-1], g['log/' + key][-1], rtol=rtol, atol=1e-6, err_msg=key)
    else:
        np.testing.assert_allclose(algo._lagrange.lagrangian_multiplier, float(g['lambda_after']), rtol=1e-6)
    _check_loss_pi_call_log(algo, g)
    _check_params(ac, g, ('actor',), 5e-5)
    _check_params(ac, g, ('reward_critic', 'cost_critic'), 2e-5)
    moved = max(float(np.abs(g[f'post/actor/{k}'] - g[f'init/actor/{k}']).max()) for k in ac.actor.state_dict())
    assert moved > 5e-3
