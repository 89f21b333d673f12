"""world_size-2 tests of the data-parallel exchange steps on CPU (gloo).  The device kernels cannot run
here, so each rank's LOCAL phase results are produced by the oracle (numpy) -- what is under test is
the communication layer (omnisafe_amd/distributed.py, Logger cross-rank statistics) and the exchange
PROTOCOL the product uses around its kernels:
  C1 gradients: clip locally, ONE flat all-reduce(SUM)/world of [3][P];  C4 advantage statistics:
  all-reduce [sum_r, sum_c, n], then [sumsq];  episode-cost mean for the Lagrange step;  C5 broadcast.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn_name, tmpdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    for p in (ROOT, os.path.join(ROOT, 'oracle')):
        if p not in sys.path:
            sys.path.insert(0, p)
    from omnisafe_amd import distributed as dist

    assert dist.init_from_env('cpu') is True
    assert dist.world_size() == world and dist.rank() == rank
    try:
        globals()[fn_name](rank, world, tmpdir)
    finally:
        torch.distributed.destroy_process_group()


def _run(fn_name, tmp_path, world=2):
    mp.spawn(_worker, args=(world, _free_port(), fn_name, str(tmp_path)), nprocs=world, join=True)


# ---------------------------------------------------------------------------------------------
def _case_grad_average(rank, world, tmpdir):
    from omnisafe_amd import distributed as dist

    P = 8448
    g = torch.Generator().manual_seed(100 + rank)
    local = torch.randn(3, P, generator=g)
    # clip-then-average (policy_gradient.py:437-442): each rank clips its own gradient first
    for net in range(3):
        n = local[net].norm()
        local[net] *= min(1.0, 40.0 / (float(n) + 1e-6))
    mine = local.clone()
    dist.all_reduce_avg_(mine)
    parts = []
    for r in range(world):
        gr = torch.Generator().manual_seed(100 + r)
        t = torch.randn(3, P, generator=gr)
        for net in range(3):
            t[net] *= min(1.0, 40.0 / (float(t[net].norm()) + 1e-6))
        parts.append(t)
    assert torch.allclose(mine, sum(parts) / world, rtol=1e-6, atol=1e-7)
    # non-contiguous views (stats[0:3]) reduce in place
    st = torch.zeros(8, dtype=torch.float64)
    st[0:3] = torch.tensor([1.0 + rank, 2.0, 3.0], dtype=torch.float64)
    dist.all_reduce_sum_(st[0:3])
    assert st[:3].tolist() == [sum(1.0 + r for r in range(world)), 2.0 * world, 3.0 * world]
    # broadcast (sync_params): rank 0's parameters win
    p = torch.full((3, P), float(rank + 1))
    dist.broadcast_(p, src=0)
    assert float(p.min()) == float(p.max()) == 1.0


def test_grad_average_and_broadcast(tmp_path):
    _run('_case_grad_average', tmp_path)


def _case_adv_stats(rank, world, tmpdir):
    """Two-phase advantage standardisation across ranks == single-process statistics over the union
    (vector_onpolicy_buffer.py:131-136 -> distributed.py:382-392)."""
    import np_oracle as O
    from omnisafe_amd import distributed as dist

    rng = np.random.default_rng(7)
    T, N = 16, 64
    adv_r = (rng.standard_normal((world, T, N)) * 3 + 0.5).astype(np.float32)
    adv_c = rng.standard_normal((world, T, N)).astype(np.float32)
    mine_r, mine_c = adv_r[rank], adv_c[rank]
    stats = torch.zeros(8, dtype=torch.float64)
    # phase 1 (device kernel osa_adv_stats_phase1 on a GPU; numpy restatement here)
    stats[0], stats[1], stats[2] = float(mine_r.astype(np.float64).sum()), float(mine_c.astype(np.float64).sum()), T * N
    dist.all_reduce_sum_(stats[0:3])
    mean_r = np.float32(np.float32(stats[0]) / np.float32(stats[2]))
    mean_c = np.float32(np.float32(stats[1]) / np.float32(stats[2]))
    # phase 2
    stats[3] = float(((mine_r - mean_r).astype(np.float32) ** 2).astype(np.float64).sum())
    dist.all_reduce_sum_(stats[3:4])
    std_r = np.float32(np.sqrt(np.float32(stats[3]) / np.float32(stats[2])))
    out_r = (mine_r - mean_r) / (std_r + np.float32(1e-8))
    # reference: statistics of the concatenation over ranks
    allr = torch.from_numpy(np.concatenate([a.reshape(-1) for a in adv_r]))
    allc = torch.from_numpy(np.concatenate([a.reshape(-1) for a in adv_c]))
    m, s = O.dist_statistics_scalar(allr)
    mc, _ = O.dist_statistics_scalar(allc)
    assert abs(float(m) - float(mean_r)) < 1e-5 and abs(float(s) - float(std_r)) < 1e-5
    assert abs(float(mc) - float(mean_c)) < 1e-6
    ref = ((torch.from_numpy(mine_r) - m) / (s + 1e-8)).numpy()
    np.testing.assert_allclose(out_r, ref, rtol=1e-5, atol=1e-6)
    assert stats[2] == world * T * N


def test_two_phase_advantage_statistics(tmp_path):
    _run('_case_adv_stats', tmp_path)


def _case_logger_stats(rank, world, tmpdir):
    """Jc for the Lagrange step = mean over all ranks' EpCost windows (logger.py:359-374); every rank
    then takes the identical lambda step (no broadcast needed)."""
    from omnisafe_amd.lagrange import Lagrange
    from omnisafe_amd.logger import Logger

    lg = Logger(tmpdir, 'exp', seed=0, verbose=False)
    lg.register_key('Metrics/EpCost', window_length=100)
    lg.register_key('Train/PolicyRatio', min_and_max=True)
    vals = [10.0, 20.0] if rank == 0 else [30.0, 40.0, 50.0]
    lg.extend('Metrics/EpCost', vals)
    lg.store({'Train/PolicyRatio': 0.9 + 0.2 * rank})
    Jc = lg.get_stats('Metrics/EpCost')[0]
    assert Jc == pytest.approx(30.0)  # (10+20+30+40+50)/5, not the mean of per-rank means
    mean, mn, mx, std = lg.get_stats('Train/PolicyRatio', True)
    assert mean == pytest.approx(1.0) and mn == pytest.approx(0.9) and mx == pytest.approx(1.1)
    assert std == pytest.approx(0.1)
    lag = Lagrange(25.0, 0.001, 0.035)
    lag.update_lagrange_multiplier(Jc)
    t = torch.tensor([lag.lagrangian_multiplier], dtype=torch.float64)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    assert float(t) == lag.lagrangian_multiplier  # identical on every rank
    lg.dump_tabular()
    lg.close()
    assert hasattr(lg, '_output_file') == (rank == 0)  # only rank 0 writes progress.csv (logger.py:120-131)


def test_logger_cross_rank_statistics(tmp_path):
    _run('_case_logger_stats', tmp_path)


def _case_seeds_and_sharding(rank, world, tmpdir):
    """Per-rank seed = cfg.seed + 1000 * rank (base_algo.py:40); steps_per_epoch is divided by
    world_size * vector_env_nums (policy_gradient.py:70-77): env batches shard with no data-path
    collective."""
    from omnisafe_amd import distributed as dist

    seed = 5 + 1000 * dist.rank()
    assert seed == 5 + 1000 * rank
    spe, n_envs = 2 * 65536, 4096
    assert spe % (dist.world_size() * n_envs) == 0
    assert spe // dist.world_size() // n_envs == 16
    t = torch.tensor([float(seed)])
    dist.all_reduce_sum_(t)
    assert float(t) == sum(5 + 1000 * r for r in range(world))


def test_seed_and_step_sharding(tmp_path):
    _run('_case_seeds_and_sharding', tmp_path)


# ---------------------------------------------------------------------------------------------
def _case_replicated_protocol(rank, world, tmpdir):
    """The replicated-data update (profiles/HISTORY.md §5): one all-gather of the epoch's rows, then every rank runs the
    WHOLE global optimiser chain on the same data -- rank r's minibatches drawn from a stream every rank can
    regenerate -- so the replicas agree without gradient traffic.  Here: the all-gather layout, the
    regenerated permutation streams and, with the oracle as the local step, bit-equal replicas that match
    the all-reduce (clip-then-average) semantics."""
    import np_oracle as O
    from omnisafe_amd import distributed as dist

    M, D_o, D_a, B, seed = 96, 12, 2, 32, 5
    g = torch.Generator().manual_seed(1000 + rank)
    local = {'obs': torch.randn(M, D_o, generator=g), 'act': torch.randn(M, D_a, generator=g),
             'logp': torch.randn(M, generator=g) - 2, 'adv_r': torch.randn(M, generator=g),
             'adv_c': torch.randn(M, generator=g), 'target_value_r': torch.randn(M, generator=g),
             'target_value_c': torch.randn(M, generator=g)}
    allr = {k: dist.all_gather_rows(v) for k, v in local.items()}
    for k, v in allr.items():  # rank r occupies rows r*M .. r*M+M-1, for every rank identically
        assert v.shape[0] == world * M and torch.equal(v[rank * M:(rank + 1) * M], local[k])
    chk = torch.stack([v.double().sum() for v in allr.values()])
    ref = chk.clone()
    dist.broadcast_(ref, src=0)
    assert torch.equal(chk, ref)
    # every rank regenerates all ranks' permutation streams (seed + 1000 r + 7919, update.py)
    perms = []
    for r in range(world):
        gen = torch.Generator().manual_seed(seed + 1000 * r + 7919)
        perms.append(torch.randperm(M, generator=gen))
    own = torch.randperm(M, generator=torch.Generator().manual_seed(seed + 1000 * rank + 7919))
    assert torch.equal(perms[rank], own)
    # global chain on the gathered data with the oracle as the per-rank gradient: identical on all ranks
    torch.manual_seed(3)
    ac = O.ActorCritic(D_o, D_a)
    params = [p for p in ac.reward_critic.parameters()]
    opt = ac.reward_critic_optimizer
    for k in range(M // B):
        grads = []
        for r in range(world):
            idx = perms[r][k * B:(k + 1) * B] + r * M
            opt.zero_grad()
            loss = torch.nn.functional.mse_loss(ac.reward_critic(allr['obs'][idx])[0] if isinstance(
                ac.reward_critic(allr['obs'][idx]), tuple) else ac.reward_critic(allr['obs'][idx]),
                allr['target_value_r'][idx])
            loss.backward()
            torch.nn.utils.clip_grad_norm_(params, 40.0)  # local clip, then average over the ranks
            grads.append([p.grad.clone() for p in params])
        for j, p in enumerate(params):
            p.grad = sum(gr[j] for gr in grads) / world
        opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    ref = flat.clone()
    dist.broadcast_(ref, src=0)
    assert torch.equal(flat, ref), 'replicas diverged'


def test_replicated_data_protocol(tmp_path):
    _run('_case_replicated_protocol', tmp_path)
