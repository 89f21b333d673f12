"""TEST INFRASTRUCTURE -- generates tests/golden/*.npz by running the UNMODIFIED reference
(/root/reference, via oracle/ref_harness.py) on seeded inputs.  Runs only in the build container
(the reference is not present on the GPU box); the .npz files are committed and travel.

    python oracle/make_golden.py            # regenerate everything

The reference code is never edited: where an intermediate value is needed (minibatch permutation,
Gaussian noise draws) the script wraps ``torch.randperm`` / ``torch.distributions.utils._standard_normal``
with recorders that return the original results untouched.
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

# (OSA_GOLDEN_OUT: another output directory -- tests/test_oracle_vs_reference.py regenerates a fixture next to the committed one)
OUT = os.environ.get('OSA_GOLDEN_OUT') or os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def _np(x):
    return x.detach().cpu().numpy().copy() if isinstance(x, torch.Tensor) else np.array(x)


# ------------------------------------------------------------------------------------------------
def gen_discount_cumsum():
    """Known-answer vectors of the reference's own unit test (tests/test_utils.py:95-115) plus the
    function evaluated on random data."""
    from omnisafe.utils.math import discount_cumsum

    out = {}
    x = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0])
    for g in (0.9, 0.99, 0.999):
        out[f'kat_{g}'] = _np(discount_cumsum(x, g))
    rng = np.random.default_rng(0)
    xr = rng.standard_normal(257).astype(np.float32)
    out['x_rand'] = xr
    out['y_rand_0.9405'] = _np(discount_cumsum(torch.from_numpy(xr.copy()), 0.99 * 0.95))
    np.savez(os.path.join(OUT, 'discount_cumsum.npz'), **out)


# ------------------------------------------------------------------------------------------------
def _ragged_schedule(rng, T, N):
    """Random path boundaries: kind 0 none, 1 terminated (bootstrap 0), 2 truncated (bootstrap v);
    last step always ends the path (epoch end)."""
    kind = rng.choice([0, 1, 2], size=(T, N), p=[0.8, 0.1, 0.1]).astype(np.int32)
    kind[-1, :] = 2
    kind[:, 0] = 0  # env 0: a single path spanning the whole epoch
    kind[-1, 0] = 2
    if N > 1:
        kind[:, 1] = 1  # env 1: every step is its own path (length-1 paths)
    return kind


def gen_buffer():
    """VectorOnPolicyBuffer.store / finish_path / get on ragged paths, all estimators."""
    from gymnasium.spaces import Box
    from omnisafe.common.buffer import VectorOnPolicyBuffer

    T, N, D_o, D_a = 24, 6, 5, 2
    rng = np.random.default_rng(1)
    inp = {
        'obs': rng.standard_normal((T, N, D_o)).astype(np.float32),
        'act': rng.standard_normal((T, N, D_a)).astype(np.float32),
        'reward': rng.standard_normal((T, N)).astype(np.float32),
        'cost': (rng.random((T, N)) < 0.3).astype(np.float32),
        'value_r': rng.standard_normal((T, N)).astype(np.float32),
        'value_c': rng.standard_normal((T, N)).astype(np.float32),
        'logp': rng.standard_normal((T, N)).astype(np.float32),
    }
    kind = _ragged_schedule(rng, T, N)
    boot_r = np.where(kind == 2, rng.standard_normal((T, N)), 0.0).astype(np.float32)
    boot_c = np.where(kind == 2, rng.standard_normal((T, N)), 0.0).astype(np.float32)
    out = dict(inp, path_end=(kind != 0).astype(np.uint8), boot_r=boot_r, boot_c=boot_c,
               gamma=0.99, lam=0.95, lam_c=0.9)
    for est in ('gae', 'gae-rtg', 'plain', 'vtrace'):
        for pc in (0.0, 0.3):
            buf = VectorOnPolicyBuffer(
                obs_space=Box(-np.inf, np.inf, (D_o,)), act_space=Box(-1, 1, (D_a,)), size=T,
                gamma=0.99, lam=0.95, lam_c=0.9, advantage_estimator=est, penalty_coefficient=pc,
                standardized_adv_r=True, standardized_adv_c=True, num_envs=N)
            for t in range(T):
                buf.store(**{k: torch.from_numpy(v[t].copy()) for k, v in inp.items()})
                for n in range(N):
                    if kind[t, n]:
                        buf.finish_path(torch.tensor([boot_r[t, n]]), torch.tensor([boot_c[t, n]]), n)
            # raw (un-standardised) per-env arrays, env-major, before get() resets pointers
            raw = {k: np.concatenate([_np(b.data[k]) for b in buf.buffers])
                   for k in ('adv_r', 'adv_c', 'target_value_r', 'target_value_c', 'discounted_ret')}
            data = buf.get()
            tag = f'{est}_pc{pc}'
            for k, v in raw.items():
                out[f'{tag}/raw/{k}'] = v
            for k, v in data.items():
                out[f'{tag}/get/{k}'] = _np(v)
    np.savez(os.path.join(OUT, 'buffer.npz'), **out)


# ------------------------------------------------------------------------------------------------
def gen_normalizer():
    from omnisafe.common.normalizer import Normalizer

    rng = np.random.default_rng(2)
    D = 7
    norm = Normalizer((D,), clip=5)
    out = {}
    batches = [rng.standard_normal((n, D)).astype(np.float32) * s + m
               for n, s, m in ((1, 1.0, 0.0), (4, 2.0, 1.0), (3, 0.5, -2.0), (16, 10.0, 0.0), (2, 1e-4, 3.0))]
    for i, b in enumerate(batches):
        y = norm.normalize(torch.from_numpy(b.copy()))
        out[f'in{i}'] = b
        out[f'out{i}'] = _np(y)
        out[f'mean{i}'] = _np(norm._mean)
        out[f'sumsq{i}'] = _np(norm._sumsq)
        out[f'var{i}'] = _np(norm._var)
        out[f'std{i}'] = _np(norm._std)
        out[f'count{i}'] = np.int64(int(norm._count))
    out['n_batches'] = len(batches)
    np.savez(os.path.join(OUT, 'normalizer.npz'), **out)


# ------------------------------------------------------------------------------------------------
class _Recorder:
    """Spy on torch.randperm and on the standard-normal draws of torch.distributions."""

    def __init__(self):
        self.perms, self.normals = [], []

    def __enter__(self):
        import torch.distributions.normal as tdn

        self._rp, self._sn, self._tdn = torch.randperm, tdn._standard_normal, tdn

        def randperm(*a, **k):
            r = self._rp(*a, **k)
            self.perms.append(r.clone())
            return r

        def std_normal(*a, **k):
            r = self._sn(*a, **k)
            self.normals.append(r.clone())
            return r

        torch.randperm = randperm
        tdn._standard_normal = std_normal
        return self

    def __exit__(self, *exc):
        torch.randperm = self._rp
        self._tdn._standard_normal = self._sn


def _state(module):
    return {k: _np(v).copy() for k, v in module.state_dict().items()}


def _make_algo(algo, env_id, n_envs, steps_per_epoch, horizon, extra_algo=None, seed=0):
    import omnisafe

    ref_harness.register_synth_env()
    d = tempfile.mkdtemp()
    cfg = {
        'seed': seed,
        'train_cfgs': {'total_steps': steps_per_epoch * 4, 'vector_env_nums': n_envs,
                       'torch_threads': 8, 'device': 'cpu'},
        'algo_cfgs': dict({'steps_per_epoch': steps_per_epoch}, **(extra_algo or {})),
        'logger_cfgs': {'use_wandb': False, 'use_tensorboard': False, 'log_dir': d},
    }
    if algo in ('PPOLag',):
        cfg['env_cfgs'] = {'horizon': horizon}
    else:  # TRPOLag.yaml / CPO.yaml have no env_cfgs key: custom env_cfgs would be rejected
        ref_harness.DEFAULT_HORIZON = horizon
    agent = omnisafe.Agent(algo, env_id, custom_cfgs=cfg)
    return agent.agent


def gen_actor_critic_step():
    """ConstraintActorCritic.step with recorded noise."""
    algo = _make_algo('PPOLag', 'SynthPointGoal1-v0', 4, 64, 8)
    ac = algo._actor_critic
    rng = np.random.default_rng(3)
    obs = rng.standard_normal((37, 60)).astype(np.float32)
    with _Recorder() as rec:
        torch.manual_seed(11)
        act, v_r, v_c, logp = ac.step(torch.from_numpy(obs.copy()))
    act_det, _, _, logp_det = ac.step(torch.from_numpy(obs.copy()), deterministic=True)
    out = {'obs': obs, 'eps': _np(rec.normals[0]), 'act': _np(act), 'value_r': _np(v_r),
           'value_c': _np(v_c), 'logp': _np(logp), 'act_det': _np(act_det),
           'logp_det': _np(logp_det)}
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in _state(getattr(ac, net)).items():
            out[f'{net}/{k}'] = v
    np.savez(os.path.join(OUT, 'actor_critic_step.npz'), **out)


def _snapshot_buffer_raw(buf):
    keys = ('obs', 'act', 'reward', 'cost', 'value_r', 'value_c', 'logp', 'adv_r', 'adv_c',
            'target_value_r', 'target_value_c', 'discounted_ret')
    return {k: np.stack([_np(b.data[k]) for b in buf.buffers], axis=1) for k in keys}  # (T, N, ...)


def gen_rollout_and_ppolag_update():
    """One reference epoch on the synthetic env: rollout (recorded env outputs + noise) -> buffer ->
    PPOLag._update (recorded permutations) -> post-update parameters."""
    N, T, horizon = 4, 40, 16
    algo = _make_algo('PPOLag', 'SynthPointGoal1-v0', N, N * T, horizon,
                      extra_algo={'update_iters': 3, 'batch_size': 64, 'kl_early_stop': False})
    ac = algo._actor_critic
    out = {'N': N, 'T': T, 'horizon': horizon}
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in _state(getattr(ac, net)).items():
            out[f'init/{net}/{k}'] = v

    # record what the raw env hands to the wrapper chain
    env_core = algo._env._env
    while hasattr(env_core, '_env'):
        env_core = env_core._env
    steps = []
    orig_step, orig_reset = env_core.step, env_core.reset

    def spy_step(action):
        r = orig_step(action)
        obs, reward, cost, term, trunc, info = r
        steps.append({'action': _np(action).copy(), 'obs': _np(obs).copy(), 'reward': _np(reward).copy(),
                      'cost': _np(cost).copy(), 'terminated': _np(term).copy(),
                      'truncated': _np(trunc).copy(),
                      'final_obs': (_np(info['final_observation']).copy()
                                    if 'final_observation' in info else np.zeros_like(_np(obs)))})
        return r

    resets = []

    def spy_reset(*a, **k):
        r = orig_reset(*a, **k)
        resets.append(_np(r[0]).copy())
        return r

    env_core.step, env_core.reset = spy_step, spy_reset
    with _Recorder() as rec:
        torch.manual_seed(5)
        algo._env.rollout(steps_per_epoch=T, agent=ac, buffer=algo._buf, logger=algo._logger)
    env_core.step, env_core.reset = orig_step, orig_reset
    out['rollout/reset_obs'] = resets[-1]
    out['rollout/resets'] = np.stack(resets)  # every reset() in call order (early termination resets mid-epoch)
    for k in steps[0]:
        out[f'rollout/{k}'] = np.stack([s[k] for s in steps])
    vec_eps = [e for e in rec.normals if e.dim() == 2]
    assert len(vec_eps) == T
    out['rollout/eps'] = np.stack([_np(e) for e in vec_eps])
    raw = _snapshot_buffer_raw(algo._buf)
    for k, v in raw.items():
        out[f'buffer/{k}'] = v
    norm = algo._env.save()['obs_normalizer']
    for k in ('_mean', '_sumsq', '_var', '_std', '_count'):
        out[f'rollout/norm{k}'] = _np(getattr(norm, k))
    out['rollout/ep_cost_window'] = np.asarray(list(algo._logger._data['Metrics/EpCost']), np.float32)
    out['rollout/ep_ret_window'] = np.asarray(list(algo._logger._data['Metrics/EpRet']), np.float32)
    out['rollout/ep_len_window'] = np.asarray(list(algo._logger._data['Metrics/EpLen']), np.float32)
    out['rollout/value_r_log_mean'] = np.float32(np.mean(algo._logger._data['Value/reward']))

    # ---- update
    out['update/lambda_before'] = _np(algo._lagrange.lagrangian_multiplier)
    out['update/Jc'] = np.float32(algo._logger.get_stats('Metrics/EpCost')[0])
    with _Recorder() as rec:
        torch.manual_seed(7)
        algo._update()
    out['update/lambda_after'] = _np(algo._lagrange.lagrangian_multiplier)
    # RandomSampler.__iter__ (torch/utils/data/sampler.py) calls randperm twice per pass: the full
    # permutation that is consumed, then one whose slice [:num_samples % n] is empty.  Keep the first.
    assert len(rec.perms) == 2 * 3
    out['update/perms'] = np.stack([_np(p) for p in rec.perms[::2]])
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in _state(getattr(ac, net)).items():
            out[f'post/{net}/{k}'] = v
    lg = algo._logger._data
    for key, name in (('Loss/Loss_pi', 'loss_pi'), ('Loss/Loss_reward_critic', 'loss_r'),
                      ('Loss/Loss_cost_critic', 'loss_c'), ('Train/PolicyRatio', 'ratio_mean'),
                      ('Train/Entropy', 'entropy'), ('Train/KL', 'kl'), ('Train/StopIter', 'stop_iter'),
                      ('Value/Adv', 'value_adv')):
        out[f'update/{name}'] = np.asarray(list(lg[key]), np.float32)
    np.savez(os.path.join(OUT, 'ppolag_epoch.npz'), **out)


def gen_trust_region_updates():
    """TRPOLag._update_actor and CPO._update_actor on a reference-collected buffer: policy gradient g,
    CG solution x, xHx, (CPO: cost gradient b, CG solution p, q/r/s, optimisation case, lambda*, nu*),
    accepted line-search index and the parameters after the step.  Intermediate tensors are captured
    by wrapping omnisafe.utils.math.conjugate_gradients as imported by the algorithm modules."""
    import omnisafe.algorithms.on_policy.base.natural_pg as npg_mod
    import omnisafe.algorithms.on_policy.base.trpo as trpo_mod
    import omnisafe.algorithms.on_policy.second_order.cpo as cpo_mod

    N, T, horizon = 4, 64, 16
    for algo_name in ('TRPOLag', 'CPO'):
        algo = _make_algo(algo_name, 'SynthPointGoal1-v0', N, N * T, horizon,
                          extra_algo={'update_iters': 2, 'batch_size': 128})
        ac = algo._actor_critic
        out = {'N': N, 'T': T}
        torch.manual_seed(21)
        algo._env.rollout(steps_per_epoch=T, agent=ac, buffer=algo._buf, logger=algo._logger)
        # make the cost signal matter: overwrite costs / adv so that the constraint is active
        if algo_name == 'TRPOLag':
            with torch.no_grad():
                algo._lagrange.lagrangian_multiplier.data.fill_(0.7)
        for net in ('actor', 'reward_critic', 'cost_critic'):
            for k, v in _state(getattr(ac, net)).items():
                out[f'init/{net}/{k}'] = v
        data = algo._buf.get()
        for k, v in data.items():
            out[f'data/{k}'] = _np(v)
        calls = []
        orig_cg = trpo_mod.conjugate_gradients

        def spy_cg(fvp, b, num_steps):
            x = orig_cg(fvp, b, num_steps)
            calls.append((b.clone(), x.clone(), fvp(x).clone()))
            return x

        for m in (npg_mod, trpo_mod, cpo_mod):
            m.conjugate_gradients = spy_cg
        ls_calls = []
        if algo_name == 'CPO':
            orig_ls = algo._cpo_search_step

            def spy_ls(**kw):
                r = orig_ls(**kw)
                ls_calls.append((kw['step_direction'].clone(), r[0].clone(), r[1], kw['optim_case'],
                                 float(kw['loss_reward_before']), float(kw['loss_cost_before'])))
                return r

            algo._cpo_search_step = spy_ls
            # CPO reads Jc from the logger window: pin it
            out['ep_cost_mean'] = np.float32(algo._logger.get_stats('Metrics/EpCost')[0])
        else:
            orig_ls = algo._search_step_size

            def spy_ls(**kw):
                r = orig_ls(**kw)
                ls_calls.append((kw['step_direction'].clone(), r[0].clone(), r[1], float(kw['loss_before'])))
                return r

            algo._search_step_size = spy_ls
            out['lambda'] = _np(algo._lagrange.lagrangian_multiplier)
        algo._update_actor(data['obs'], data['act'], data['logp'], data['adv_r'], data['adv_c'])
        for m in (npg_mod, trpo_mod, cpo_mod):
            m.conjugate_gradients = orig_cg
        out['g'] = _np(calls[0][0])
        out['x'] = _np(calls[0][1])
        out['Fx'] = _np(calls[0][2])  # includes damping
        if algo_name == 'CPO':
            out['b'] = _np(calls[1][0])
            out['p'] = _np(calls[1][1])
            out['optim_case'] = np.int32(ls_calls[0][3])
            out['loss_reward_before'] = np.float32(ls_calls[0][4])
            out['loss_cost_before'] = np.float32(ls_calls[0][5])
        else:
            out['loss_before'] = np.float32(ls_calls[0][3])
        out['step_direction'] = _np(ls_calls[0][0])
        out['final_step'] = _np(ls_calls[0][1])
        out['accept_step'] = np.int32(ls_calls[0][2])
        for k, v in _state(ac.actor).items():
            out[f'post/actor/{k}'] = v
        lg = algo._logger._data
        for key in ('Misc/Alpha', 'Misc/xHx', 'Misc/H_inv_g', 'Misc/gradient_norm', 'Misc/FinalStepNorm',
                    'Train/KL'):
            if key in lg and len(lg[key]):
                out['log/' + key] = np.asarray(list(lg[key]), np.float32)
        if algo_name == 'CPO':
            for key in ('Misc/Lambda_star', 'Misc/Nu_star', 'Misc/A', 'Misc/B', 'Misc/q', 'Misc/r', 'Misc/s',
                        'Misc/cost_gradient_norm'):
                out['log/' + key] = np.asarray(list(lg[key]), np.float32)
        np.savez(os.path.join(OUT, f'{algo_name.lower()}_actor_update.npz'), **out)


# (algorithm, tag, extra algo_cfgs, extra lagrange_cfgs): the sibling on-policy algorithms that only
# override the surrogate / multiplier logic
SIBLINGS = [
    ('PolicyGradient', 'policygradient', {}, None),
    ('PPO', 'ppo', {}, None),
    ('NaturalPG', 'naturalpg', {}, None),
    ('TRPO', 'trpo', {}, None),
    ('PDO', 'pdo', {}, {'cost_limit': 1.0}),
    ('RCPO', 'rcpo', {}, {'cost_limit': 1.0}),
    ('IPO', 'ipo', {'cost_limit': 8.0, 'kappa': 0.5}, None),
    ('OnCRPO', 'oncrpo_reward', {'cost_limit': 1000.0}, None),
    ('OnCRPO', 'oncrpo_cost', {'cost_limit': 0.0, 'distance': 0.1}, None),
    ('CPPOPID', 'cppopid', {}, {'cost_limit': 1.0}),
    ('TRPOPID', 'trpopid', {}, {'cost_limit': 1.0}),
    ('PCPO', 'pcpo', {'cost_limit': 1.0}, None),
    ('FOCOPS', 'focops', {'focops_eta': 0.02}, {'cost_limit': 1.0}),
    ('FOCOPS', 'focops_masked', {'focops_eta': 1e-4}, {'cost_limit': 1.0}),  # trust mask partially active
    ('CUP', 'cup', {}, {'cost_limit': 1.0}),
    ('P3O', 'p3o', {'cost_limit': 1.0, 'kappa': 2.0}, None),
    # KL early stop (policy_gradient.py:391-397): 4 passes allowed, the threshold is crossed after the second
    ('PPOLag', 'ppolag_earlystop', {'kl_early_stop': True, 'target_kl': 3e-4, 'update_iters': 4},
     {'cost_limit': 1.0}),
]


def _gen_update_golden(algo_name, env_id, out_name, N, T, horizon, extra, lag, update_iters=2, model=None):
    """One reference `_update()` of `algo_name` on a reference-collected buffer of N x T transitions of
    `env_id`: inputs (initial parameters, `buf.get()` output, EpCost window, recorded permutations) and
    outputs (parameters of all three networks, multiplier, logged statistics) -> tests/golden/<out_name>."""
    import omnisafe
    from omnisafe.utils.config import get_default_kwargs_yaml

    base = get_default_kwargs_yaml(algo_name, env_id, 'on-policy').todict()
    trust_region = 'cg_iters' in base['algo_cfgs']
    ea = dict({'update_iters': update_iters, 'batch_size': 128 if trust_region else 64, 'kl_early_stop': False},
              **extra)
    ref_harness.register_synth_env()
    ref_harness.DEFAULT_HORIZON = horizon
    d = tempfile.mkdtemp()
    cfg = {'seed': 0,
           'train_cfgs': {'total_steps': N * T * 4, 'vector_env_nums': N, 'torch_threads': 8, 'device': 'cpu'},
           'algo_cfgs': dict({'steps_per_epoch': N * T}, **ea),
           'logger_cfgs': {'use_wandb': False, 'use_tensorboard': False, 'log_dir': d}}
    if lag:
        cfg['lagrange_cfgs'] = lag
    if model:  # network shapes other than the YAML's [64, 64] (utils/model.py:73-111 builds any hidden_sizes)
        cfg['model_cfgs'] = model
    algo = omnisafe.Agent(algo_name, env_id, custom_cfgs=cfg).agent
    ac = algo._actor_critic
    out = {'N': N, 'T': T, 'algo': algo_name, 'env_id': env_id}
    torch.manual_seed(31)
    algo._env.rollout(steps_per_epoch=T, agent=ac, buffer=algo._buf, logger=algo._logger)
    out['ep_cost_window'] = np.asarray(list(algo._logger._data['Metrics/EpCost']), np.float32)
    out['Jc'] = np.float32(algo._logger.get_stats('Metrics/EpCost')[0])
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in _state(getattr(ac, net)).items():
            out[f'init/{net}/{k}'] = v
    captured = {}
    orig_get = algo._buf.get

    def spy_get():
        r = orig_get()
        if not captured:  # CUP calls get() twice (the buffer is empty the second time around)
            captured.update({k: v.clone() for k, v in r.items()})
        return {k: v.clone() for k, v in captured.items()}

    algo._buf.get = spy_get
    if hasattr(algo, '_lagrange'):
        lm = algo._lagrange.lagrangian_multiplier
        out['lambda_before'] = np.float32(float(lm))
    with _Recorder() as rec:
        torch.manual_seed(33)
        algo._update()
    for k, v in captured.items():
        out[f'data/{k}'] = _np(v)
    if hasattr(algo, '_lagrange'):
        out['lambda_after'] = np.float32(float(algo._lagrange.lagrangian_multiplier))
    # RandomSampler draws two permutations per pass (see gen_rollout_and_ppolag_update)
    out['perms'] = np.stack([_np(p) for p in rec.perms[::2]]) if rec.perms else np.zeros((0, N * T), np.int64)
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in _state(getattr(ac, net)).items():
            out[f'post/{net}/{k}'] = v
    for key, val in algo._logger._data.items():
        if key.startswith(('Loss/', 'Train/', 'Misc/', 'Metrics/LagrangeMultiplier', 'Value/Adv')):
            vals = list(val) if not isinstance(val, (int, float)) else [val]
            if len(vals) and all(isinstance(x, (int, float, np.floating)) for x in vals):
                out['log/' + key] = np.asarray(vals, np.float32)
    np.savez(os.path.join(OUT, out_name), **out)
    return out


def gen_sibling_updates(only=None):
    """One reference `_update()` per sibling algorithm (SafetyPointGoal1 shapes, M = 160)
    -> tests/golden/sibling_<tag>.npz."""
    N, T, horizon = 4, 40, 16
    for algo_name, tag, extra, lag in SIBLINGS:
        if only and tag not in only:
            continue
        out = _gen_update_golden(algo_name, 'SynthPointGoal1-v0', f'sibling_{tag}.npz', N, T, horizon, extra, lag)
        print('sibling', tag, {k: out[k] for k in ('Jc', 'lambda_before', 'lambda_after') if k in out},
              'perms', out['perms'].shape)


# (tag, algorithm, env id, extra algo_cfgs, extra lagrange_cfgs): one whole reference `_update()` at the
# observation / action shapes of every BASELINE.json config, M = 16 x 256 = 4096 transitions.
#   config 2  PPOLag  SafetyPointGoal1 60/2      config 3  CPO     SafetyCarGoal1 72/2
#   config 4  PPOLag  SafetyHumanoidVelocity 376/17 (wide input)
#   config 5  TRPOLag SafetyAntVelocity 27/8 (rows not 16-byte aligned: the padding path; D_a = 8)
# The multiplier starts at 0.5 so that the cost advantage carries weight in the surrogate; CPO's limit
# sits just below the rollout's mean episode cost so that the constraint is active.
CONFIG_SHAPES = [
    ('config2_ppolag_point', 'PPOLag', 'SynthPointGoal1-v0', {}, {'lagrangian_multiplier_init': 0.5, 'cost_limit': 0.5}),
    ('config3_cpo_car', 'CPO', 'SynthCarGoal1-v0', {'cost_limit': 0.71}, None),
    # the infeasible-recovery branch (case 0: a pure step along -F^-1 b), half of CPO's updates early in training
    ('config3_cpo_car_case0', 'CPO', 'SynthCarGoal1-v0', {'cost_limit': 0.5}, None),
    ('config4_ppolag_humanoid', 'PPOLag', 'SynthHumanoid-v0', {},
     {'lagrangian_multiplier_init': 0.5, 'cost_limit': 0.5}),
    ('config5_trpolag_ant', 'TRPOLag', 'SynthAnt-v0', {}, {'lagrangian_multiplier_init': 0.5, 'cost_limit': 0.5}),
]


# One whole reference `_update()` on networks OUTSIDE the YAML's [64, 64] family (the layer-wise path of
# csrc/general_mlp.hip incl. its skinny kernels for 64-row minibatches): unequal widths, actor != critics.
_M256 = {'actor': {'hidden_sizes': [256, 128]}, 'critic': {'hidden_sizes': [256, 128]}}
HIDDEN_SHAPES = [
    ('hidden256x128_ppolag_point', 'PPOLag', 'SynthPointGoal1-v0', {},
     {'lagrangian_multiplier_init': 0.5, 'cost_limit': 0.5}, _M256),
    ('hidden256x128_trpolag_ant', 'TRPOLag', 'SynthAnt-v0', {}, {'lagrangian_multiplier_init': 0.5, 'cost_limit': 0.5},
     _M256),
    ('hidden256x128_focops_point', 'FOCOPS', 'SynthPointGoal1-v0', {'focops_eta': 0.02}, {'cost_limit': 1.0}, _M256),
    ('hidden96x40x24_p3o_point', 'P3O', 'SynthPointGoal1-v0', {'cost_limit': 0.5, 'kappa': 2.0}, None,
     {'actor': {'hidden_sizes': [96, 40, 24]}, 'critic': {'hidden_sizes': [80, 48]}}),
]


# Generated only when named (`make_golden.py hidden-shapes hidden1024x1024_ppolag_point`) and never committed (26 MB):
# the network shape of the reference's one published timing table (docs/source/start/efficiency.rst:15-23), recorded
# LIVE on the GPU box's host from the staged archive by tests/test_general_mlp_gpu.py::test_update_of_the_live_reference_at_1024x1024
_M1024 = {'actor': {'hidden_sizes': [1024, 1024]}, 'critic': {'hidden_sizes': [1024, 1024]}}
LIVE_HIDDEN_SHAPES = [
    ('hidden1024x1024_ppolag_point', 'PPOLag', 'SynthPointGoal1-v0', {},
     {'lagrangian_multiplier_init': 0.5, 'cost_limit': 0.5}, _M1024),
]


def gen_hidden_shape_updates(only=None):
    N, T, horizon = 16, 64, 16
    for tag, algo_name, env_id, extra, lag, model in HIDDEN_SHAPES + LIVE_HIDDEN_SHAPES:
        if (only and tag not in only) or (not only and tag.startswith('hidden1024')):
            continue
        out = _gen_update_golden(algo_name, env_id, f'{tag}.npz', N, T, horizon, extra, lag, model=model)
        print(tag, {k: out[k] for k in ('Jc', 'lambda_before', 'lambda_after') if k in out}, 'perms',
              out['perms'].shape, out['init/actor/mean.0.weight'].shape)


def gen_config_shape_updates(only=None):
    N, T, horizon = 16, 256, 16
    for tag, algo_name, env_id, extra, lag in CONFIG_SHAPES:
        if only and tag not in only:
            continue
        out = _gen_update_golden(algo_name, env_id, f'{tag}.npz', N, T, horizon, extra, lag)
        print(tag, {k: out[k] for k in ('Jc', 'lambda_before', 'lambda_after') if k in out}, 'perms',
              out['perms'].shape, {k: out[k] for k in out if k.startswith('log/Misc')})


# ------------------------------------------------------------------------------------------------
# data parallelism: the UNMODIFIED reference with TWO ranks (train_cfgs.parallel = 2, gloo on the CPU)
#
# The reference joins a process group when MASTER_ADDR is set (utils/distributed.py:83-104, called from
# algo_wrapper.py:152), gives rank r the seed cfg.seed + 1000 r (base_algo.py:40), divides steps_per_epoch by the world
# size (policy_gradient.py:73-77), broadcasts rank 0's parameters (:98-99, utils/distributed.py:201-226), standardises
# the advantages with global statistics (vector_onpolicy_buffer.py:131-136 -> utils/distributed.py:382-392), averages the
# window mean of EpCost for the Lagrange step (logger.py:359-374), clips every network's gradient LOCALLY and then
# averages it over the ranks before each optimiser step (policy_gradient.py:437-442, 478-483, 519-524 ->
# utils/distributed.py:167-198), averages the KL (:390), the Fisher-vector products (natural_pg.py:112) and the
# line-search quantities (trpo.py:114-118, 181-185).  One `_update()` per config, everything a rank sees recorded.
# (tag, algorithm, env id, N envs per rank, T steps per rank, extra algo_cfgs, lagrange_cfgs)
DP2_CONFIGS = [
    # BASELINE config 2's shapes under 2 ranks: 2 x 32 optimiser steps of 64 rows per rank
    ('dp2_ppolag_point', 'PPOLag', 'SynthPointGoal1-v0', 16, 128, {},
     {'lagrangian_multiplier_init': 0.5, 'cost_limit': 0.5}),
    # BASELINE config 4 (8-rank config: PPOLag on 376 / 17)
    ('dp2_ppolag_humanoid', 'PPOLag', 'SynthHumanoid-v0', 16, 128, {},
     {'lagrangian_multiplier_init': 0.5, 'cost_limit': 0.5}),
    # BASELINE config 5 (8-rank config: TRPOLag on 27 / 8; FVP / line-search averages + batch-128 critic passes)
    ('dp2_trpolag_ant', 'TRPOLag', 'SynthAnt-v0', 16, 128, {},
     {'lagrangian_multiplier_init': 0.5, 'cost_limit': 0.5}),
    # the large-batch setting (PPOLag.yaml's GPU-env blocks: batch_size 8192, update_iters 8) under 2 ranks:
    # 2 passes x 2 steps of 2048 rows per rank
    ('dp2_ppolag_point_largebatch', 'PPOLag', 'SynthPointGoal1-v0', 16, 256, {'batch_size': 2048},
     {'lagrangian_multiplier_init': 0.5, 'cost_limit': 0.5}),
    # BASELINE config 3 (8-rank config: CPO on 72 / 2): reward AND cost gradients, both conjugate-gradient solves and
    # the two-constraint line search on rank averages (second_order/cpo.py:57-462); cost_limit below the synthetic
    # env's episode cost so that the constraint is active (c > 0: not the trivial TRPO case)
    ('dp2_cpo_car', 'CPO', 'SynthCarGoal1-v0', 16, 128, {'cost_limit': 0.5}, None),
]


def _dp2_worker(rank, world, port, spec, tmp):
    tag, algo_name, env_id, N, T, extra, lag = spec
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    os.environ.pop('IN_DIST', None)
    torch.set_num_threads(2)
    ref_harness.import_reference()
    import omnisafe
    from omnisafe.utils import distributed as rdist
    from omnisafe.utils.config import get_default_kwargs_yaml

    base = get_default_kwargs_yaml(algo_name, env_id, 'on-policy').todict()
    trust_region = 'cg_iters' in base['algo_cfgs']
    ea = dict({'update_iters': 2, 'batch_size': 128 if trust_region else 64, 'kl_early_stop': False}, **extra)
    ref_harness.register_synth_env()
    ref_harness.DEFAULT_HORIZON = 16
    spe = world * N * T  # the GLOBAL steps_per_epoch: every rank collects spe / world / N = T vector steps
    cfg = {'seed': 0,
           'train_cfgs': {'total_steps': spe * 4, 'vector_env_nums': N, 'torch_threads': 2, 'device': 'cpu',
                          'parallel': world},
           'algo_cfgs': dict({'steps_per_epoch': spe}, **ea),
           'logger_cfgs': {'use_wandb': False, 'use_tensorboard': False, 'log_dir': os.path.join(tmp, f'log{rank}')}}
    if lag is not None:
        cfg['lagrange_cfgs'] = lag
    algo = omnisafe.Agent(algo_name, env_id, custom_cfgs=cfg).agent  # (fork() joins the gloo group)
    assert rdist.world_size() == world and rdist.get_rank() == rank
    assert algo._steps_per_epoch == T and algo._seed == 1000 * rank
    ac = algo._actor_critic
    out = {'N': N, 'T': T, 'world': world, 'seed': 0}
    for net in ('actor', 'reward_critic', 'cost_critic'):  # after sync_params: rank 0's parameters everywhere
        for k, v in _state(getattr(ac, net)).items():
            out[f'init/{net}/{k}'] = v
    torch.manual_seed(31 + 1000 * rank)
    algo._env.rollout(steps_per_epoch=T, agent=ac, buffer=algo._buf, logger=algo._logger)
    raw = _snapshot_buffer_raw(algo._buf)
    out['raw/adv_r'], out['raw/adv_c'] = raw['adv_r'], raw['adv_c']  # (T, N), before the global standardisation
    out['ep_cost_window'] = np.asarray(list(algo._logger._data['Metrics/EpCost']), np.float32)
    out['Jc'] = np.float32(algo._logger.get_stats('Metrics/EpCost')[0])  # cross-rank mean (a collective: all ranks)
    captured = {}
    orig_get = algo._buf.get

    def spy_get():
        r = orig_get()
        captured.update({k: v.clone() for k, v in r.items()})
        return r

    algo._buf.get = spy_get
    has_lag = hasattr(algo, '_lagrange')
    out['lambda_before'] = np.float32(float(algo._lagrange.lagrangian_multiplier) if has_lag else np.nan)
    with _Recorder() as rec:
        torch.manual_seed(33 + 1000 * rank)
        algo._update()
    for k, v in captured.items():
        out[f'data/{k}'] = _np(v)
    out['lambda_after'] = np.float32(float(algo._lagrange.lagrangian_multiplier) if has_lag else np.nan)
    out['perms'] = np.stack([_np(p) for p in rec.perms[::2]])  # RandomSampler draws two permutations per pass
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in _state(getattr(ac, net)).items():
            out[f'post/{net}/{k}'] = v
    for key, val in algo._logger._data.items():
        if key.startswith(('Loss/', 'Train/', 'Misc/', 'Metrics/LagrangeMultiplier', 'Value/Adv')):
            vals = list(val) if not isinstance(val, (int, float)) else [val]
            if len(vals) and all(isinstance(x, (int, float, np.floating)) for x in vals):
                out['log/' + key] = np.asarray(vals, np.float32)
    np.savez(os.path.join(tmp, f'rank{rank}.npz'), **out)
    import torch.distributed as tdist

    tdist.barrier()
    tdist.destroy_process_group()


def gen_dp2_updates(only=None, world=2):
    """tests/golden/dp2_<tag>.npz: what every rank of a 2-rank run of the unmodified reference fed into and got out
    of one `_update()`.  Rank-specific arrays under `r<rank>/...` (data = the rank's `buf.get()` output, perms, raw
    advantages, EpCost window, logged per-rank statistics); `init/`, `post/`, `Jc`, `lambda_*` are identical on all
    ranks (asserted here) and stored once."""
    import socket

    import torch.multiprocessing as mp

    # `tag@T` in `only`: the same config with T vector steps per rank (the 376-wide recording at 8 ranks is 25 MB at T = 128)
    t_of = {o.split('@')[0]: int(o.split('@')[1]) for o in (only or []) if '@' in o}
    only = [o.split('@')[0] for o in only] if only else only
    for spec in DP2_CONFIGS:
        tag = spec[0]
        if only and tag not in only:
            continue
        if tag in t_of:
            spec = spec[:4] + (t_of[tag],) + spec[5:]
        tmp = tempfile.mkdtemp()
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        mp.spawn(_dp2_worker, args=(world, port, spec, tmp), nprocs=world, join=True)
        parts = [dict(np.load(os.path.join(tmp, f'rank{r}.npz'))) for r in range(world)]
        out = {}
        for k, v in parts[0].items():
            if k.startswith(('init/', 'post/')) or k in ('N', 'T', 'world', 'seed', 'Jc', 'lambda_before', 'lambda_after'):
                for r in range(1, world):
                    assert np.array_equal(v, parts[r][k], equal_nan=True), (tag, k, 'differs between ranks')
                out[k] = v
        for r in range(world):
            for k, v in parts[r].items():
                if k.startswith(('data/', 'raw/', 'log/')) or k in ('perms', 'ep_cost_window'):
                    out[f'r{r}/{k}'] = v
        out['algo'], out['env_id'] = spec[1], spec[2]
        tag = tag.replace('dp2_', f'dp{world}_')
        np.savez_compressed(os.path.join(OUT, f'{tag}.npz'), **out)
        moved = max(float(np.abs(out[k] - out['init/' + k[5:]]).max()) for k in out if k.startswith('post/actor/'))
        print(tag, 'Jc', out['Jc'], 'lambda', out['lambda_before'], '->', out['lambda_after'], 'perms',
              out['r0/perms'].shape, 'actor moved by', moved, os.path.getsize(os.path.join(OUT, f'{tag}.npz')), 'B')



def _record_rollout(algo, T, seed):
    """Run the reference adapter's rollout with spies on the raw env and on the policy noise; returns the
    recorded trace + resulting buffer / normaliser / episode-metric contents (keys as in ppolag_epoch)."""
    ac = algo._actor_critic
    out = {}
    env_core = algo._env._env
    while hasattr(env_core, '_env'):
        env_core = env_core._env
    steps, resets = [], []
    orig_step, orig_reset = env_core.step, env_core.reset

    def spy_step(action):
        r = orig_step(action)
        obs, reward, cost, term, trunc, info = r
        steps.append({'action': _np(action).copy(), 'obs': _np(obs).copy(), 'reward': _np(reward).copy(),
                      'cost': _np(cost).copy(), 'terminated': _np(term).copy(),
                      'truncated': _np(trunc).copy(),
                      'final_obs': (_np(info['final_observation']).copy()
                                    if 'final_observation' in info else np.zeros_like(_np(obs)))})
        return r

    def spy_reset(*a, **k):
        r = orig_reset(*a, **k)
        resets.append(_np(r[0]).copy())
        return r

    env_core.step, env_core.reset = spy_step, spy_reset
    with _Recorder() as rec:
        torch.manual_seed(seed)
        algo._env.rollout(steps_per_epoch=T, agent=ac, buffer=algo._buf, logger=algo._logger)
    env_core.step, env_core.reset = orig_step, orig_reset
    out['rollout/reset_obs'] = resets[-1]
    out['rollout/resets'] = np.stack(resets)  # every reset() in call order (early termination resets mid-epoch)
    for k in steps[0]:
        out[f'rollout/{k}'] = np.stack([s[k] for s in steps])
    vec_eps = [e for e in rec.normals if e.dim() == 2]
    assert len(vec_eps) == T
    out['rollout/eps'] = np.stack([_np(e) for e in vec_eps])
    for k, v in _snapshot_buffer_raw(algo._buf).items():
        out[f'buffer/{k}'] = v
    norm = algo._env.save()['obs_normalizer']
    for k in ('_mean', '_sumsq', '_var', '_std', '_count'):
        out[f'rollout/norm{k}'] = _np(getattr(norm, k))
    for key, name in (('Metrics/EpCost', 'ep_cost_window'), ('Metrics/EpRet', 'ep_ret_window'),
                      ('Metrics/EpLen', 'ep_len_window'), ('Metrics/EpBudget', 'ep_budget_window')):
        if key in algo._logger._data:
            out[f'rollout/{name}'] = np.asarray([float(x) for x in algo._logger._data[key]], np.float32)
    return out


def gen_saute_simmer():
    """SauteAdapter / SimmerAdapter rollouts of the reference on the synthetic env (budget small enough
    that the unsafe branch is reached) and the Simmer PID budget controller's trajectory."""
    import omnisafe

    N, T, horizon = 4, 40, 16
    for algo_name, tag, extra in (
            ('PPOSaute', 'saute', {'safety_budget': 1.0, 'max_ep_len': 16, 'unsafe_reward': -0.5}),
            ('PPOSimmerPID', 'simmer', {'safety_budget': 1.0, 'upper_budget': 2.0, 'max_ep_len': 16,
                                        'unsafe_reward': -0.5})):
        ref_harness.register_synth_env()
        ref_harness.DEFAULT_HORIZON = horizon
        d = tempfile.mkdtemp()
        cfg = {'seed': 0,
               'train_cfgs': {'total_steps': N * T * 4, 'vector_env_nums': N, 'torch_threads': 8, 'device': 'cpu'},
               'algo_cfgs': dict({'steps_per_epoch': N * T, 'update_iters': 2}, **extra),
               'logger_cfgs': {'use_wandb': False, 'use_tensorboard': False, 'log_dir': d}}
        if tag == 'simmer':
            cfg['control_cfgs'] = {'kp': 0.05, 'ki': 0.01, 'kd': 0.02, 'polyak': 0.9}
        algo = omnisafe.Agent(algo_name, 'SynthPointGoal1-v0', custom_cfgs=cfg).agent
        out = {'N': N, 'T': T, 'horizon': horizon}
        for net in ('actor', 'reward_critic', 'cost_critic'):
            for k, v in _state(getattr(algo._actor_critic, net)).items():
                out[f'init/{net}/{k}'] = v
        out.update(_record_rollout(algo, T, seed=5))
        if tag == 'simmer':
            traj = []
            for jc in (1.5, 0.2, 3.0, 0.9):
                algo._env.control_budget(torch.as_tensor(jc, dtype=torch.float32))
                traj.append(np.concatenate([_np(algo._env._safety_budget).reshape(-1)[:1],
                                            _np(algo._env._rel_safety_budget).reshape(-1)[:1]]))
            out['control/jc'] = np.asarray((1.5, 0.2, 3.0, 0.9), np.float32)
            out['control/budget_rel'] = np.stack(traj)
        np.savez(os.path.join(OUT, f'{tag}_rollout.npz'), **out)
        print(tag, 'unsafe steps:', int((out['buffer/reward'] == np.float32(-0.5)).sum()),
              'EpBudget', out.get('rollout/ep_budget_window'))


def gen_early_terminated():
    """EarlyTerminatedAdapter rollout of the reference (single env, as it requires): 400 steps on the
    synthetic env with cost_limit 1.5, so that episodes end both by the time limit (every 16 steps) and
    by early termination (second unit of accumulated cost; the accumulator survives time-limit resets)."""
    import omnisafe

    N, T, horizon = 1, 400, 16
    ref_harness.register_synth_env()
    ref_harness.DEFAULT_HORIZON = horizon  # PPOEarlyTerminated.yaml has no env_cfgs key
    d = tempfile.mkdtemp()
    cfg = {'seed': 0,
           'train_cfgs': {'total_steps': N * T * 4, 'vector_env_nums': N, 'torch_threads': 1, 'device': 'cpu'},
           'algo_cfgs': {'steps_per_epoch': N * T, 'update_iters': 2, 'cost_limit': 1.5},
           'logger_cfgs': {'use_wandb': False, 'use_tensorboard': False, 'log_dir': d}}
    algo = omnisafe.Agent('PPOEarlyTerminated', 'SynthPointGoal1-v0', custom_cfgs=cfg).agent
    out = {'N': N, 'T': T, 'horizon': horizon, 'cost_limit': np.float32(1.5)}
    for net in ('actor', 'reward_critic', 'cost_critic'):
        for k, v in _state(getattr(algo._actor_critic, net)).items():
            out[f'init/{net}/{k}'] = v
    rec = _record_rollout(algo, T, seed=7)
    for k, v in rec.items():  # a single reference env returns unbatched tensors: restore the env axis
        if k in ('rollout/action', 'rollout/obs', 'rollout/final_obs'):
            v = v.reshape(T, N, -1)
        elif k in ('rollout/reward', 'rollout/cost', 'rollout/terminated', 'rollout/truncated'):
            v = v.reshape(T, N)
        elif k == 'rollout/reset_obs':
            v = v.reshape(N, -1)
        elif k == 'rollout/resets':
            v = v.reshape(v.shape[0], N, -1)
        elif k == 'rollout/eps':
            v = v.reshape(T, N, -1)
        out[k] = v
    np.savez(os.path.join(OUT, 'early_terminated_rollout.npz'), **out)
    print('early terminated: resets', out['rollout/resets'].shape[0] - 1, 'truncations',
          int(out['rollout/truncated'].sum()), 'episodes', len(out['rollout/ep_len_window']),
          'zeroed rewards', int((out['buffer/reward'] == 0).sum()))


def gen_config_defaults():
    """Snapshot of the `defaults` blocks of the reference's on-policy YAML files."""
    import json

    import yaml

    cfg_dir = os.path.join(ref_harness.REFERENCE_ROOT, 'omnisafe', 'configs', 'on-policy')
    out = {}
    for name in sorted(os.listdir(cfg_dir)):
        if name.endswith('.yaml'):
            out[name[:-5]] = yaml.safe_load(open(os.path.join(cfg_dir, name)))['defaults']
    json.dump(out, open(os.path.join(OUT, 'config_defaults.json'), 'w'), indent=1, sort_keys=True)

LEARNING_CFG = {
    # shared by the reference runs below and tests/test_learning_gpu.py
    'env_id': 'SynthReach-v0', 'epochs': 10, 'vector_env_nums': 16, 'steps_per_epoch': 4096,
    'horizon': 50, 'cost_limit': 2.0, 'tail_epochs': 3,
}


LEARNING_ALGOS = ('PPOLag', 'TRPOLag', 'CPO', 'PolicyGradient', 'PPO', 'NaturalPG', 'TRPO', 'PDO', 'RCPO', 'CPPOPID',
                  'TRPOPID', 'PCPO', 'FOCOPS', 'CUP', 'IPO', 'P3O', 'OnCRPO', 'PPOSaute', 'TRPOSaute',
                  'PPOSimmerPID', 'TRPOSimmerPID')


def learning_custom_cfgs(algo, seed, device, log_dir, defaults, c=None):
    """custom_cfgs of one learning-curve run, derived from the algorithm's YAML defaults so that the cost
    limit lands where that algorithm reads it.  tests/test_learning_gpu.py keeps a copy of this function
    (tests may not import this generator on the GPU box) and feeds it omnisafe_amd's defaults, which a CPU
    test pins to the reference's YAML files.  No env_cfgs: several YAMLs have no such key and would reject
    it; the env's default horizon (50) applies."""
    c = c or LEARNING_CFG
    cfg = {
        'seed': seed,
        'train_cfgs': {'total_steps': c['steps_per_epoch'] * c['epochs'],
                       'vector_env_nums': c['vector_env_nums'], 'device': device},
        'algo_cfgs': {'steps_per_epoch': c['steps_per_epoch']},
        'logger_cfgs': {'log_dir': log_dir, 'save_model_freq': 1000},
    }
    if 'cost_limit' in defaults.get('lagrange_cfgs', {}):
        cfg['lagrange_cfgs'] = {'cost_limit': c['cost_limit']}
    if 'cost_limit' in defaults['algo_cfgs']:
        cfg['algo_cfgs']['cost_limit'] = c['cost_limit']
    if 'safety_budget' in defaults['algo_cfgs']:  # Saute / Simmer: budget = the limit, episodes of 50 steps
        cfg['algo_cfgs'].update({'safety_budget': c['cost_limit'], 'max_ep_len': c['horizon']})
        if 'upper_budget' in defaults['algo_cfgs']:
            cfg['algo_cfgs']['upper_budget'] = 2 * c['cost_limit']
    return cfg


def gen_learning_curves(algos=LEARNING_ALGOS, seeds=tuple(range(20)), merge=True, part=None):
    """Train the unmodified reference on the learnable point-reach CMDP and record the per-epoch
    Metrics/EpRet, EpCost (and LagrangeMultiplier) of every seed (progress.csv of the reference's
    logger): the comparison target for "episode return/cost within +-1 sigma over 3 seeds"."""
    import csv
    import json

    import omnisafe

    ref_harness.register_reach_env()
    c = LEARNING_CFG
    # part: write a separate file (tools: several single-thread runs in parallel, merged afterwards
    # by merge_learning_parts)
    path = os.path.join(OUT, 'learning_reach.json' if part is None else f'_learning_part_{part}.json')
    out = {'config': dict(c), 'curves': {}}
    if merge and os.path.exists(path):
        out['curves'] = json.load(open(path))['curves']
    for algo in algos:
        out['curves'].setdefault(algo, {})
        for seed in seeds:
            d = tempfile.mkdtemp()
            from omnisafe.utils.config import get_default_kwargs_yaml

            defaults = get_default_kwargs_yaml(algo, c['env_id'], 'on-policy').todict()
            cfg = learning_custom_cfgs(algo, seed, 'cpu', d, defaults)
            cfg['train_cfgs']['torch_threads'] = 1
            cfg['logger_cfgs'].update({'use_wandb': False, 'use_tensorboard': False})
            omnisafe.Agent(algo, c['env_id'], custom_cfgs=cfg).learn()
            rows = None
            for root, _, files in os.walk(d):
                if 'progress.csv' in files:
                    rows = list(csv.DictReader(open(os.path.join(root, 'progress.csv'))))
            keys = ['EpRet', 'EpCost'] + (['LagrangeMultiplier'] if 'Metrics/LagrangeMultiplier' in rows[0] else [])
            out['curves'][algo][str(seed)] = {k: [float(r[f'Metrics/{k}']) for r in rows] for k in keys}
            json.dump(out, open(path, 'w'), indent=1, sort_keys=True)


def merge_learning_parts():
    import glob
    import json

    out = {'config': dict(LEARNING_CFG), 'curves': {}}
    for f in sorted(glob.glob(os.path.join(OUT, '_learning_part_*.json'))):
        for algo, seeds in json.load(open(f))['curves'].items():
            out['curves'].setdefault(algo, {}).update(seeds)
        os.remove(f)
    out['curves'] = {a: dict(sorted(s.items(), key=lambda kv: int(kv[0]))) for a, s in out['curves'].items()}
    json.dump(out, open(os.path.join(OUT, 'learning_reach.json'), 'w'), sort_keys=True, separators=(',', ':'))


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_harness.import_reference()
    torch.set_num_threads(1)  # deterministic reductions
    gen_discount_cumsum()
    gen_buffer()
    gen_normalizer()
    gen_actor_critic_step()
    gen_rollout_and_ppolag_update()
    gen_trust_region_updates()
    gen_sibling_updates()
    gen_config_shape_updates()
    gen_hidden_shape_updates()
    gen_dp2_updates()
    gen_saute_simmer()
    gen_early_terminated()
    gen_config_defaults()
    gen_learning_curves()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    # `make_golden.py`                         everything, serially (the 420 learning runs take ~8 CPU-hours)
    # `make_golden.py learning ALGO SEED`      one learning run -> tests/golden/_learning_part_ALGO_SEED.json
    #                                          (how the committed fixture was made: 8 of these in parallel,
    #                                          OMP_NUM_THREADS=1 each)
    # `make_golden.py dp2 [TAG ...]`           the 2-rank (gloo) reference runs -> tests/golden/dp2_<tag>.npz
    # `make_golden.py merge-learning`          fold the part files into tests/golden/learning_reach.json
    if len(sys.argv) >= 4 and sys.argv[1] == 'learning':
        ref_harness.import_reference()
        gen_learning_curves(algos=(sys.argv[2],), seeds=(int(sys.argv[3]),), merge=False,
                            part=f'{sys.argv[2]}_{sys.argv[3]}')
    elif len(sys.argv) >= 2 and sys.argv[1] == 'config-shapes':
        ref_harness.import_reference()
        torch.set_num_threads(1)
        gen_config_shape_updates(only=sys.argv[2:] or None)
    elif len(sys.argv) >= 2 and sys.argv[1] == 'hidden-shapes':
        ref_harness.import_reference()
        torch.set_num_threads(1)
        gen_hidden_shape_updates(only=sys.argv[2:] or None)
    elif len(sys.argv) >= 2 and sys.argv[1] == 'dp2':
        gen_dp2_updates(only=sys.argv[2:] or None)
    elif len(sys.argv) >= 2 and sys.argv[1] == 'dp4':  # the same recordings with FOUR ranks -> tests/golden/dp4_<...>.npz
        gen_dp2_updates(only=sys.argv[2:] or ['dp2_ppolag_point', 'dp2_trpolag_ant'], world=4)
    elif len(sys.argv) >= 2 and sys.argv[1] == 'dp8':  # EIGHT ranks (the world size BASELINE.json quotes): dp8_<...>.npz
        # (committed set: `dp8` [= dp2_ppolag_point], `dp8 dp2_ppolag_humanoid@32 dp2_trpolag_ant dp2_cpo_car`:
        # BASELINE configs 4, 5, 3 at the world size they are quoted on)
        gen_dp2_updates(only=sys.argv[2:] or ['dp2_ppolag_point'], world=8)
    elif len(sys.argv) >= 2 and sys.argv[1] == 'merge-learning':
        if os.path.exists(os.path.join(OUT, 'learning_reach.json')):  # keep what is already there
            os.replace(os.path.join(OUT, 'learning_reach.json'), os.path.join(OUT, '_learning_part_0prev.json'))
        merge_learning_parts()
    else:
        main()
