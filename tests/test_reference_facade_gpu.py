"""SURVEY.md 8(b) acceptance run: the REFERENCE's own facade drives the plugin end to end on the GPU.

    import omnisafe                      # the unmodified reference (staged archive, oracle/stage_reference.py)
    omnisafe_amd.install()               # registry swap (omnisafe_amd/plugin.py)
    omnisafe.Agent(algo, env, custom_cfgs={'train_cfgs': {'device': 'cuda:0', ...}}).learn()

i.e. the reference's AlgoWrapper (algo_wrapper.py:56-184) builds its Config from the reference's YAML
files, runs its own checks and looks the algorithm up in its registry -- and gets the HIP classes.  Checked:

  * the agent object is omnisafe_amd's class, `learn()` returns finite (ep_ret, ep_cost, ep_len);
  * `progress.csv` has exactly the columns (names and order) of a run of the unmodified reference class on
    the CPU with the same configuration (policy_gradient.py:133-236, logger.py:284-319);
  * every `torch_save/epoch-N.pt` is loaded by the reference's `Evaluator` (evaluator.py:153-303, through
    the reference's own `Agent.evaluate`, algo_wrapper.py:221-236), whose rebuilt actor holds the device
    actor's parameters bit for bit and plays episodes on the CPU twin of the env.

The GPU box has no /root/reference: the package comes from oracle/_ref/omnisafe_ref.zip, staged by
`__graft_entry__.build()` in the build container.  Env: SynthReach-v0 (episodes of 50 steps in both
implementations, so T = 50 rows per env hold whole episodes and the Lagrange update sees finite costs).
"""
import csv
import glob
import os

import numpy as np
import pytest
import torch

import ref_harness

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref_harness.reference_available(),
                                 reason='no reference: neither /root/reference nor oracle/_ref/omnisafe_ref.zip '
                                        '(run `python __graft_entry__.py` in the build container)')]

N_ENVS, T, EPOCHS = 16, 50, 2


def _cfg(device, log_dir, defaults):
    cfg = {'seed': 3,
           'train_cfgs': {'device': device, 'total_steps': N_ENVS * T * EPOCHS, 'vector_env_nums': N_ENVS,
                          'torch_threads': 1},
           'algo_cfgs': {'steps_per_epoch': N_ENVS * T, 'update_iters': 2},
           'logger_cfgs': {'use_wandb': False, 'use_tensorboard': False, 'log_dir': log_dir,
                           'save_model_freq': 1}}
    # cost limit of 2 where the algorithm reads it (lagrange_cfgs for the *Lag family, algo_cfgs for CPO)
    if 'cost_limit' in defaults.get('lagrange_cfgs', {}):
        cfg['lagrange_cfgs'] = {'cost_limit': 2.0}
    if 'cost_limit' in defaults['algo_cfgs']:
        cfg['algo_cfgs']['cost_limit'] = 2.0
    return cfg


def _progress(log_root):
    files = glob.glob(os.path.join(log_root, '**', 'progress.csv'), recursive=True)
    assert len(files) == 1, files
    rows = list(csv.reader(open(files[0])))
    return os.path.dirname(files[0]), rows[0], rows[1:]


@pytest.fixture(scope='module')
def omnisafe_ref():
    omnisafe = ref_harness.import_reference()
    ref_harness.register_reach_env()  # CPU twin of SynthReach-v0, registered with the reference's env registry
    return omnisafe


@pytest.mark.parametrize('algo', ['PPOLag', 'TRPOLag', 'CPO'])
def test_reference_agent_runs_the_plugin(omnisafe_ref, tmp_path, algo):
    omnisafe = omnisafe_ref
    import omnisafe_amd
    from omnisafe.algorithms import registry as ref_registry
    from omnisafe.utils.config import get_default_kwargs_yaml

    defaults = get_default_kwargs_yaml(algo, 'SynthReach-v0', 'on-policy').todict()

    # ---- the unmodified reference class on the CPU: the csv header to match
    omnisafe_amd.uninstall()
    ref_cls = ref_registry.REGISTRY.get(algo)
    assert ref_cls.__module__.startswith('omnisafe.')
    ref_dir = str(tmp_path / 'ref')
    ref_agent = omnisafe.Agent(algo, 'SynthReach-v0', custom_cfgs=_cfg('cpu', ref_dir, defaults))
    assert type(ref_agent.agent) is ref_cls
    ref_ret = ref_agent.learn()
    _, ref_header, ref_rows = _progress(ref_dir)
    assert len(ref_rows) == EPOCHS

    # ---- the same call with the plugin installed and device cuda:0
    try:
        swapped = omnisafe_amd.install()
        assert algo in swapped
        amd_dir = str(tmp_path / 'amd')
        agent = omnisafe.Agent(algo, 'SynthReach-v0', custom_cfgs=_cfg('cuda:0', amd_dir, defaults))
        # what the registry hands out: a class named like the reference's, subclass of the reference's class
        # (isinstance checks written against the reference hold) with the HIP implementation first in the MRO
        assert type(agent.agent).__mro__[1] is omnisafe_amd.algorithms.registry.get(algo)
        assert isinstance(agent.agent, ref_cls) and type(agent.agent).__name__ == algo
        assert type(agent.agent).__module__.startswith('omnisafe_amd.')
        assert type(agent.agent).learn.__module__.startswith('omnisafe_amd.')
        ep_ret, ep_cost, ep_len = agent.learn()
        assert np.isfinite([ep_ret, ep_cost]).all() and ep_len == 50.0 == ref_ret[2]
        run_dir, header, rows = _progress(amd_dir)
        assert header == ref_header, (set(header) ^ set(ref_header))
        assert len(rows) == EPOCHS
        vals = {k: float(v) for k, v in zip(header, rows[-1])}
        assert vals['TotalEnvSteps'] == N_ENVS * T * EPOCHS and vals['Train/Epoch'] == EPOCHS - 1
        assert np.isfinite([v for k, v in vals.items() if not k.endswith('/Delta')]).all()

        # ---- checkpoints through the reference's Evaluator (its Agent.evaluate walks torch_save/*.pt)
        saved = sorted(os.listdir(os.path.join(run_dir, 'torch_save')))
        assert saved == [f'epoch-{i}.pt' for i in range(EPOCHS + 1)]
        agent.evaluate(num_episodes=1)  # algo_wrapper.py:221-236 -> evaluator.py:366-397,153-303,399-490
        ev = agent._evaluator  # noqa: SLF001  (the reference's object: holds the last checkpoint it loaded)
        last = ev._model_name  # noqa: SLF001
        ref_actor = ev._actor  # noqa: SLF001  reference GaussianLearningActor rebuilt by ActorBuilder
        ck = torch.load(os.path.join(run_dir, 'torch_save', last), weights_only=False)
        assert set(ck) == {'pi', 'obs_normalizer'}
        assert set(ck['obs_normalizer']) == {'_mean', '_sumsq', '_var', '_std', '_count', '_clip'}
        for k, v in ref_actor.state_dict().items():
            assert torch.equal(v, ck['pi'][k]), k
        if last == f'epoch-{EPOCHS}.pt':  # the final checkpoint = the parameters now on the device
            for k, v in agent.agent._actor_critic.actor.state_dict().items():  # noqa: SLF001
                assert torch.equal(v.cpu(), ck['pi'][k]), k
        # a standalone Evaluator as `omnisafe eval` would build it, deterministic episodes on the CPU twin
        ev2 = omnisafe.Evaluator()
        ev2.load_saved(save_dir=run_dir, model_name=f'epoch-{EPOCHS}.pt')
        rets, costs = ev2.evaluate(num_episodes=2)
        assert len(rets) == 2 and np.isfinite(rets).all() and np.isfinite(costs).all()
        obs = torch.randn(5, 60)
        with torch.no_grad():
            mean_ref = ev2._actor.predict(obs, deterministic=True)  # noqa: SLF001
        mean_dev = agent.agent._actor_critic.step(obs.to('cuda:0'), deterministic=True)[0]  # noqa: SLF001
        np.testing.assert_allclose(mean_dev.cpu().numpy(), mean_ref.numpy(), rtol=1e-4, atol=2e-6)
    finally:
        omnisafe_amd.uninstall()
    assert ref_registry.REGISTRY.get(algo) is ref_cls
