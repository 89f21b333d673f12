"""GPU: the data-parallel path (world_size 2) end to end.  Two ranks share the single GPU of the test
box and talk over gloo (OSA_DIST_BACKEND=gloo; RCCL refuses two ranks on one device), which exercises
exactly the code the RCCL runs execute: per-rank env shards and seeds, broadcast of the initial
parameters, two-phase advantage statistics, per-step clip -> flat gradient all-reduce -> Adam, KL
all-reduce, FVP / line-search all-reduces, cross-rank logger statistics."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, algo, tmpdir, dp_mode='replicated'):
    os.environ.update(OSA_DP_MODE=dp_mode, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), OSA_DIST_BACKEND='gloo', OSA_SINGLE_DEVICE_RANKS='1',
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    sys.path.insert(0, ROOT)
    import omnisafe_amd
    from omnisafe_amd import distributed as dist

    cfg = {'seed': 4, 'train_cfgs': {'device': 'cuda:0', 'total_steps': 2 * 2 * 64 * 8, 'vector_env_nums': 64},
           'algo_cfgs': {'steps_per_epoch': 2 * 64 * 8, 'update_iters': 2, 'batch_size': 64},
           'logger_cfgs': {'log_dir': tmpdir, 'verbose': False}, 'env_cfgs': {'horizon': 4, 'cost_p': 0.3}}
    agent = omnisafe_amd.Agent(algo, 'SynthTiny-v0', custom_cfgs=cfg)
    a = agent.agent
    assert dist.world_size() == world and a._steps_per_epoch == 8        # 1024 / (2 ranks * 64 envs)
    assert a._seed == 4 + 1000 * rank
    p0 = a._actor_critic.params.clone()
    chk = p0.clone()
    dist.broadcast_(chk, src=0)
    assert torch.equal(chk, p0), 'sync_params: ranks must start from rank 0 parameters'
    ep_ret, ep_cost, ep_len = agent.learn()
    p = a._actor_critic.params
    lo, hi = p.clone(), p.clone()
    torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
    torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
    assert torch.equal(lo, hi), 'replicas diverged'
    assert torch.isfinite(p).all() and not torch.equal(p, p0)
    assert ep_len == 4.0 and 0.3 < ep_cost < 2.5
    m = a._actor_critic.adam_step.cpu().tolist()
    if algo == 'PPOLag':
        assert m == [2 * 2 * 8] * 3  # 2 epochs x 2 passes x 8 minibatches (512 local rows / 64)
    if rank == 0:
        torch.save(p.cpu(), os.path.join(tmpdir, f'params_{algo}_{dp_mode}.pt'))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('algo', ['PPOLag', 'TRPOLag', 'CPO'])
def test_two_ranks_on_one_gpu(tmp_path, algo):
    mp.spawn(_worker, args=(2, _free_port(), algo, str(tmp_path)), nprocs=2, join=True)


def test_replicated_and_allreduce_dp_modes_agree(tmp_path):
    """The two data-parallel implementations -- per-step flat gradient all-reduce vs all-gathered rollout
    with the whole global step computed on every rank -- are the same algorithm: same rollouts (same
    seeds), same Lagrange steps; only the minibatch permutations differ in how they are drawn, so
    compare distributions, and check the replicated mode against the all-reduce semantics exactly in
    tests/test_mlp_gpu.py::test_replicated_data_parallel_step_equals_allreduce_semantics."""
    for mode in ('replicated', 'allreduce'):
        mp.spawn(_worker, args=(2, _free_port(), 'PPOLag', str(tmp_path), mode), nprocs=2, join=True)
    a = torch.load(os.path.join(str(tmp_path), 'params_PPOLag_replicated.pt'))
    b = torch.load(os.path.join(str(tmp_path), 'params_PPOLag_allreduce.pt'))
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    # same initialisation, same data, 32 Adam steps of lr 3e-4 with different minibatch orders:
    # parameters stay within a few lr-steps of each other
    assert float((a - b).abs().max()) < 32 * 3e-4 * 2


def _worker_shape(rank, world, port, algo, env_id, tmpdir, dp_mode, want_path, want_chunked):
    """Two ranks at the shapes of the 8-GPU BASELINE configs (4: PPOLag on 376 / 17; 5: TRPOLag on 27 / 8 with its
    batch-128 critic passes): the update must run on the cooperative persistent data-parallel passes of round 3
    (`replicated-wide-split` = osa_ppo_split_dp_pass; chunked = osa_ppo_dp_chunked_pass), not on per-step launches."""
    os.environ.update(OSA_DP_MODE=dp_mode, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), OSA_DIST_BACKEND='gloo',
                      OSA_SINGLE_DEVICE_RANKS='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    sys.path.insert(0, ROOT)
    import omnisafe_amd
    from omnisafe_amd import distributed as dist

    cfg = {'seed': 4, 'train_cfgs': {'device': 'cuda:0', 'total_steps': 2 * 2 * 64 * 8, 'vector_env_nums': 64},
           'algo_cfgs': {'steps_per_epoch': 2 * 64 * 8, 'update_iters': 2},
           'logger_cfgs': {'log_dir': tmpdir, 'verbose': False}, 'env_cfgs': {'horizon': 4, 'cost_p': 0.3}}
    agent = omnisafe_amd.Agent(algo, env_id, custom_cfgs=cfg)
    a = agent.agent
    assert dist.world_size() == world and a._steps_per_epoch == 8
    p0 = a._actor_critic.params.clone()
    agent.learn()
    p = a._actor_critic.params
    lo, hi = p.clone(), p.clone()
    torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
    torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
    assert torch.equal(lo, hi), 'replicas diverged'
    assert torch.isfinite(p).all() and not torch.equal(p, p0)
    up = a._updater
    assert up.last_path == want_path, up.last_path
    if want_chunked is not None:
        assert up._dp.get('chunked') is want_chunked
    if rank == 0:
        torch.save(p.cpu(), os.path.join(tmpdir, f'params_{algo}_{dp_mode}.pt'))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('algo,env_id,want_path,want_chunked,steps', [
    ('PPOLag', 'SynthHumanoid-v0', 'replicated-wide-split', None, 32),  # BASELINE config 4: 376 / 17, B = 64
    ('TRPOLag', 'SynthAnt-v0', 'replicated', True, 16),                 # BASELINE config 5: 27 / 8, critics B = 128
])
def test_two_ranks_at_the_8gpu_config_shapes(tmp_path, algo, env_id, want_path, want_chunked, steps):
    """The cooperative persistent DP passes against the per-step RCCL-style path (`allreduce`: gradient kernel ->
    flat all-reduce -> Adam per optimiser step): same rollouts, same Lagrange steps, minibatch orders drawn
    differently -> parameters within a few learning-rate steps of each other (exact step-for-step equivalence:
    tests/test_mlp_gpu.py::test_wide_split_data_parallel_... / ::test_chunked_data_parallel_...)."""
    mp.spawn(_worker_shape, args=(2, _free_port(), algo, env_id, str(tmp_path), 'replicated', want_path,
                                  want_chunked), nprocs=2, join=True)
    mp.spawn(_worker_shape, args=(2, _free_port(), algo, env_id, str(tmp_path), 'allreduce', 'per-step', None),
             nprocs=2, join=True)
    a = torch.load(os.path.join(str(tmp_path), f'params_{algo}_replicated.pt'))
    b = torch.load(os.path.join(str(tmp_path), f'params_{algo}_allreduce.pt'))
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    lr = 1e-3 if algo == 'TRPOLag' else 3e-4
    # (the trust-region actor step is the same on both sides up to the FVP / line-search all-reduces' rounding;
    # the critics take `steps` Adam steps with different minibatch orders)
    assert float((a[1:] - b[1:]).abs().max()) < steps * lr * 2
