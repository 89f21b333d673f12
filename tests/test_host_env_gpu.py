"""north_star: "OnPolicyAdapter.rollout feeds vectorized Safety-Gymnasium envs" -- envs that live on the HOST and
are registered with the REFERENCE's env registry (`@env_register`, omnisafe/envs/core.py:300-421), not with this
package.  With the plugin installed, `omnisafe.Agent(algo, env_id, device cuda:0)` must build such an env through
the reference's `envs.core.make`, honour `need_time_limit_wrapper` / `need_auto_reset_wrapper`
(adapter/online_adapter.py:120-132) and drive it through `omnisafe_amd.host_env.HostEnvBridge` (one D2H / H2D pair
per vector step) while the rest of the step stays on the device.

  * 'Test-v0': the reference's own test double of a user env (tests/simple_env.py:30-90; single env, both wrappers,
    Python `random` rewards) -- staged with the reference archive by `__graft_entry__.build()`;
  * 'HostReach-v0': the CPU twin of SynthReach-v0 (oracle/ref_harness.py, numpy dynamics of oracle/np_oracle.py)
    registered under an id this package does NOT own, as a vector env with gymnasium's auto-reset convention.

Row-level parity of the bridged rollout against the reference's recording: tests/test_rollout_gpu.py::
test_rollout_on_reference_trace[host-env-bridge].  Staging logic without a GPU: tests/test_host_logic.py.
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
                                 reason='no reference: neither /root/reference nor oracle/_ref/omnisafe_ref.zip')]


@pytest.fixture(scope='module')
def omnisafe_ref():
    omnisafe = ref_harness.import_reference()
    ref_harness.import_simple_env()  # registers 'Test-v0' with the reference's registry
    reach_cls = ref_harness.register_reach_env()
    from omnisafe.envs.core import env_register, support_envs

    if 'HostReach-v0' not in support_envs():
        @env_register
        class HostReachEnv(reach_cls):  # same dynamics, an id omnisafe_amd's own registry does not know
            _support_envs = ['HostReach-v0']
    return omnisafe


def _progress(log_root):
    files = glob.glob(os.path.join(log_root, '**', 'progress.csv'), recursive=True)
    assert len(files) == 1, files
    rows = list(csv.reader(open(files[0])))
    return rows[0], rows[1:]


CASES = {
    # env id: (vector_env_nums, steps_per_epoch, epochs, episode length or None)
    'Test-v0': (1, 60, 2, None),
    'HostReach-v0': (8, 8 * 50, 2, 50.0),
}


@pytest.mark.parametrize('env_id', list(CASES))
def test_plugin_trains_on_reference_registered_host_env(omnisafe_ref, tmp_path, env_id):
    omnisafe = omnisafe_ref
    import omnisafe_amd
    from omnisafe_amd import envs as amd_envs
    from omnisafe_amd.host_env import AutoReset, HostEnvBridge, TimeLimit

    assert env_id not in amd_envs.ENV_REGISTRY
    n_envs, spe, epochs, ep_len = CASES[env_id]

    def cfg(device, log_dir):
        return {'seed': 1,
                'train_cfgs': {'device': device, 'total_steps': spe * epochs, 'vector_env_nums': n_envs,
                               'torch_threads': 1},
                'algo_cfgs': {'steps_per_epoch': spe, 'update_iters': 2, 'batch_size': 32},
                'logger_cfgs': {'use_wandb': False, 'use_tensorboard': False, 'log_dir': log_dir,
                                'save_model_freq': 1},
                'lagrange_cfgs': {'cost_limit': 2.0}}

    # the unmodified reference on the CPU: the csv header to match
    omnisafe_amd.uninstall()
    ref_dir = str(tmp_path / 'ref')
    ref_agent = omnisafe.Agent('PPOLag', env_id, custom_cfgs=cfg('cpu', ref_dir))
    ref_ret = ref_agent.learn()
    ref_header, ref_rows = _progress(ref_dir)
    assert len(ref_rows) == epochs
    try:
        assert 'PPOLag' in omnisafe_amd.install()
        amd_dir = str(tmp_path / 'amd')
        agent = omnisafe.Agent('PPOLag', env_id, custom_cfgs=cfg('cuda:0', amd_dir))
        assert type(agent.agent).__module__.startswith('omnisafe_amd.')
        bridge = agent.agent._env._env  # noqa: SLF001
        assert isinstance(bridge, HostEnvBridge)
        if env_id == 'Test-v0':  # wrappers on the host side of the boundary, as online_adapter.py:120-132 orders them
            assert isinstance(bridge.host_env, AutoReset) and isinstance(bridge.host_env._env, TimeLimit)  # noqa: SLF001
        ep_ret, ep_cost, ep_len_out = agent.learn()
        assert np.isfinite([ep_ret, ep_cost, ep_len_out]).all()
        if ep_len is not None:
            assert ep_len_out == ep_len == ref_ret[2]
        header, rows = _progress(amd_dir)
        assert header == ref_header, (set(header) ^ set(ref_header))
        assert len(rows) == epochs
        vals = {k: float(v) for k, v in zip(header, rows[-1])}
        assert vals['TotalEnvSteps'] == spe * epochs
        assert np.isfinite([v for k, v in vals.items() if not k.endswith('/Delta')]).all()
        # everything but the env ran on the device, through the HIP library
        ac = agent.agent._actor_critic  # noqa: SLF001
        assert ac.params.is_cuda and agent.agent._buf.data['obs'].is_cuda  # noqa: SLF001
        # PCIe traffic of the boundary: 4 D_a down, 4 (D_o + 5) up (+ 4 D_o when an env finished) per env-step
        d_o, d_a = int(bridge.observation_space.shape[0]), int(bridge.action_space.shape[0])
        down, up = bridge.pcie_bytes_per_env_step()
        assert down == 4 * d_a
        assert 4 * (d_o + 5) <= up <= 4 * (2 * d_o + 5) + 4 * d_o
        assert bridge.steps == spe * epochs // n_envs
    finally:
        omnisafe_amd.uninstall()


def test_unknown_env_id_is_a_key_error():
    from omnisafe_amd import envs as amd_envs

    with pytest.raises(KeyError, match='NoSuchEnv-v0'):
        amd_envs.make('NoSuchEnv-v0', num_envs=2, device='cuda:0')


def test_bridge_moves_one_copy_each_way_per_step():
    """The device side of a bridged step: the action leaves through ONE pinned D2H copy, the env outputs arrive
    through ONE pinned H2D copy (head only when no env finished), and the returned tensors are device views."""
    from omnisafe_amd.host_env import HostEnvBridge
    from test_host_logic import _HostVecEnv

    host, twin = _HostVecEnv(n=64, d_o=12, d_a=3, horizon=4), _HostVecEnv(n=64, d_o=12, d_a=3, horizon=4)
    br = HostEnvBridge(host, 'cuda:0')
    assert br._up_h.is_pinned() and br._act_h.is_pinned()  # noqa: SLF001
    o, _ = br.reset()
    assert o.is_cuda and torch.equal(o.cpu(), twin.reset()[0])
    rng = np.random.default_rng(0)
    for t in range(9):
        act = torch.from_numpy(rng.standard_normal((64, 3)).astype(np.float32))
        obs, r, c, term, trunc, info = br.step(act.cuda())
        eo, er, ec, _, etrunc, einfo = twin.step(act)
        for a, b in ((obs, eo), (r, er), (c, ec), (trunc, etrunc.float())):
            assert a.is_cuda and torch.equal(a.cpu(), b)
        assert torch.equal(host.actions[-1], act)  # bit-exact float32 down
        assert ('final_observation' in info) == ('final_observation' in einfo)
        if 'final_observation' in einfo:
            m = torch.from_numpy(einfo['_final_observation'])
            assert torch.equal(info['final_observation'].cpu()[m], einfo['final_observation'][m])
            assert torch.equal(info['_final_observation'].cpu(), m.float())
