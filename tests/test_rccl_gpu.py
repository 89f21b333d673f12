"""GPU: the `nccl` (= RCCL on ROCm) backend for real, on the single GPU of the test box.

The reference selects `nccl` whenever the device is a GPU (omnisafe/utils/distributed.py:75-80,100) and
all-reduces gradients per parameter tensor (:167-198).  omnisafe_amd/distributed.py selects the same
backend; the multi-rank tests (tests/test_dp_gpu.py, tests/test_distributed_gloo.py) must use gloo
because RCCL refuses two ranks on one device.  Here a world of ONE rank with
OSA_DIST_FORCE_COLLECTIVES=1 initialises the RCCL process group and then issues every collective of the
multi-GPU path on device tensors -- `all_reduce` (flat gradients, fp64 statistics),
`all_gather_into_tensor` (the per-epoch rollout gather of the replicated mode), `broadcast` (initial
parameters) -- and runs whole PPOLag / TRPOLag / CPO epochs in each data-parallel mode on top of it.
It also measures the per-call latency floor of each collective and the per-optimiser-step time of the
`allreduce` mode (gradient kernel -> RCCL all-reduce -> Adam), written to
gpurun_out/rccl_world1_timing.json (copied to profiles/ by hand).  No 1 -> 8 GPU curve can be measured on
this box; a world of one only proves the calls, their arguments and their ordering on the stream.
"""
import json
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


def _env(port, dp_mode):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                      OSA_DIST_FORCE_COLLECTIVES='1', OSA_DP_MODE=dp_mode, HSA_ENABLE_IPC_MODE_LEGACY='0')
    os.environ.pop('OSA_DIST_BACKEND', None)
    sys.path.insert(0, ROOT)


def _time_us(fn, iters=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def _collectives_worker(_rank, port, out_path):
    _env(port, 'allreduce')
    from omnisafe_amd import distributed as dist

    assert dist.init_from_env('cuda:0') is False  # a world of one: "not distributed" for the caller
    assert torch.distributed.is_initialized() and torch.distributed.get_backend() == 'nccl'
    assert dist.world_size() == 1 and dist.rank() == 0 and dist.collectives_active()
    dev = torch.device('cuda:0')
    res = {'backend': torch.distributed.get_backend(), 'world_size': 1}
    # C1: the flat gradient message of one optimiser step (3 networks x 8448 padded floats, config 2)
    g = torch.randn(3, 8448, device=dev)
    want = g.clone()
    dist.all_reduce_avg_(g)
    assert torch.equal(g, want)
    res['all_reduce_grads_101KB_us'] = _time_us(lambda: dist.all_reduce_avg_(g))
    # Humanoid-sized message (3 x 29 696 floats)
    gh = torch.randn(3, 29696, device=dev)
    res['all_reduce_grads_356KB_us'] = _time_us(lambda: dist.all_reduce_avg_(gh))
    # C4: fp64 advantage statistics, a slice of the stats vector
    stats = torch.arange(8, dtype=torch.float64, device=dev)
    dist.all_reduce_sum_(stats[0:3])
    assert stats.tolist() == list(range(8))
    res['all_reduce_stats_24B_us'] = _time_us(lambda: dist.all_reduce_sum_(stats[0:3]))
    # non-contiguous input goes through a contiguous staging copy
    nc = torch.randn(6, 4, device=dev)[:, 1]
    want = nc.clone()
    dist.all_reduce_sum_(nc)
    assert torch.equal(nc, want)
    # the per-epoch rollout gather of the replicated mode: all_gather_into_tensor
    obs = torch.randn(65536, 60, device=dev)
    out = torch.empty(65536, 60, device=dev)
    got = dist.all_gather_rows(obs, out=out)
    assert got is out and torch.equal(out, obs)
    res['all_gather_obs_15.7MB_us'] = _time_us(lambda: dist.all_gather_rows(obs, out=out), iters=50)
    assert torch.equal(dist.all_gather_rows(obs[:100]), obs[:100])
    # C5: initial parameters
    p = torch.randn(3, 8448, device=dev)
    want = p.clone()
    dist.broadcast_(p, src=0)
    assert torch.equal(p, want)
    res['broadcast_params_101KB_us'] = _time_us(lambda: dist.broadcast_(p, src=0))
    dist.barrier()
    # ---- the `allreduce` data-parallel mode on config-2 shapes: per-step gradient kernel -> ONE flat RCCL
    # all-reduce -> Adam (the reference's structure with 1 message per step instead of 19)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_mlp_gpu import make_ac
    from omnisafe_amd.update import PPOUpdater

    torch.manual_seed(3)
    M, B = 65536, 64
    ac = make_ac(60, 2)
    data = {'obs': torch.randn(M, 60, device=dev), 'act': torch.randn(M, 2, device=dev),
            'logp': torch.randn(M, device=dev) * 0.1 - 2.8, 'target_value_r': torch.randn(M, device=dev),
            'target_value_c': torch.randn(M, device=dev), 'adv_r': torch.randn(M, device=dev),
            'adv_c': torch.randn(M, device=dev)}
    lam = torch.tensor([0.2], device=dev)
    p_init, g = ac.params.clone(), torch.Generator().manual_seed(11)
    perms = [[torch.randperm(M, generator=g)] for _ in range(3)]

    def reset():
        ac.params.copy_(p_init)
        ac.adam_m.zero_()
        ac.adam_v.zero_()
        ac.adam_step.zero_()

    after = {}
    for mode in ('allreduce-eager', 'allreduce', 'replicated', 'replicated-steps'):
        reset()
        if mode == 'allreduce-eager':
            os.environ['OSA_UPDATE_GRAPH'] = '0'
        up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False,
                        dp_mode='allreduce' if mode == 'allreduce-eager' else mode)
        for k in range(3):  # eager warm-up pass (LDS attributes) / capture / replay: the third one is timed
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out_ = up.run(data, lam, perms=perms[k] if mode.startswith('allreduce') else None, actor_lr=3e-4,
                          critic_lr=3e-4)
            b.record()
            torch.cuda.synchronize()
        os.environ.pop('OSA_UPDATE_GRAPH', None)
        assert out_['steps'] == M // B and bool(torch.isfinite(ac.params).all())
        # round 5: the per-step all-reduce mode is ONE captured hipGraph per pass incl. its 1024 RCCL all-reduces
        assert up.last_path == {'allreduce-eager': 'per-step', 'allreduce': 'allreduce-graph'}.get(mode, 'replicated')
        after[mode] = ac.params.clone()
        res[f'dp_mode_{mode}_us_per_step_world1'] = a.elapsed_time(b) * 1e3 / out_['steps']
    # the captured pass performs the eager launches' arithmetic: same bits after three updates
    assert torch.equal(after['allreduce'], after['allreduce-eager'])
    assert float((after['allreduce'] - p_init).abs().max()) > 1e-3
    torch.distributed.destroy_process_group()
    with open(out_path, 'w') as f:
        json.dump(res, f, indent=1)


def test_rccl_world1_collectives_and_dp_modes(tmp_path):
    out = str(tmp_path / 'rccl.json')
    mp.spawn(_collectives_worker, args=(_free_port(), out), nprocs=1, join=True)
    res = json.load(open(out))
    assert res['backend'] == 'nccl'
    print('RCCL world-1 timings:', json.dumps(res))
    dst = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(dst, exist_ok=True)
    json.dump(res, open(os.path.join(dst, 'rccl_world1_timing.json'), 'w'), indent=1)


def _agent_worker(_rank, port, algo, dp_mode, tmpdir):
    _env(port, dp_mode)
    import omnisafe_amd
    from omnisafe_amd import distributed as dist

    cfg = {'seed': 4, 'train_cfgs': {'device': 'cuda:0', 'total_steps': 2 * 128 * 16, 'vector_env_nums': 128},
           'algo_cfgs': {'steps_per_epoch': 128 * 16, 'update_iters': 2},
           'logger_cfgs': {'log_dir': tmpdir, 'verbose': False}, 'env_cfgs': {'horizon': 8, 'cost_p': 0.3}}
    agent = omnisafe_amd.Agent(algo, 'SynthPointGoal1-v0', custom_cfgs=cfg)
    assert torch.distributed.get_backend() == 'nccl' and dist.collectives_active()
    p0 = agent.agent._actor_critic.params.clone()
    ep_ret, ep_cost, ep_len = agent.learn()
    p = agent.agent._actor_critic.params
    assert bool(torch.isfinite(p).all()) and not torch.equal(p, p0)
    assert ep_len == 8.0 and 1.0 < ep_cost < 4.0
    if algo == 'PPOLag':
        # (2 passes per epoch: the first pass of the first epoch is eager, every later one a replayed graph)
        assert agent.agent._updater.last_path == ('allreduce-graph' if dp_mode == 'allreduce' else 'replicated')
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('algo,dp_mode', [('PPOLag', 'allreduce'), ('PPOLag', 'replicated'),
                                          ('PPOLag', 'replicated-steps'), ('TRPOLag', 'allreduce'),
                                          ('CPO', 'allreduce')])
def test_agents_over_rccl_world1(tmp_path, algo, dp_mode):
    mp.spawn(_agent_worker, args=(_free_port(), algo, dp_mode, str(tmp_path)), nprocs=1, join=True)


def _large_batch_graph_worker(_rank, port, out_path):
    """Round 4: the data-parallel form of the LARGE-BATCH update (B >= 2048) -- per optimiser step: partial gradients ->
    slab reduce + LOCAL clip -> ONE flat RCCL all-reduce (average) -> Adam (policy_gradient.py:437-443 order), the whole
    pass incl. its collectives captured in ONE hipGraph and replayed (`dp-large-batch-graph`).  World of one rank over
    the real `nccl` backend: proves the capture of RCCL collectives, the stream ordering and the arithmetic (an average
    over one rank is the identity: the result must equal the single-process step) and measures the per-step cost of
    the path without wire time."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_mlp_gpu import make_ac
    from omnisafe_amd.update import PPOUpdater

    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    M, B, iters = 65536, 16384, 8
    data = {'obs': torch.randn(M, 60, device=dev), 'act': torch.randn(M, 2, device=dev),
            'logp': torch.randn(M, device=dev) * 0.1 - 2.8, 'target_value_r': torch.randn(M, device=dev),
            'target_value_c': torch.randn(M, device=dev), 'adv_r': torch.randn(M, device=dev),
            'adv_c': torch.randn(M, device=dev)}
    lam = torch.tensor([0.2], device=dev)
    g = torch.Generator().manual_seed(9)
    perms = [[torch.randperm(M, generator=g) for _ in range(iters)] for _ in range(3)]

    def run3(ac, up):
        """three updates (eager warm-up pass / capture / replays); returns us per optimiser step of the third"""
        t = None
        for k in range(3):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = up.run(data, lam, perms=perms[k], actor_lr=3e-4, critic_lr=3e-4)
            b.record()
            torch.cuda.synchronize()
            t = a.elapsed_time(b) * 1e3 / out['steps']
        return t

    # ---- single process (no process group yet): the reference point
    ac0 = make_ac(60, 2)
    p_init = ac0.params.clone()
    up0 = PPOUpdater(ac0, batch_size=B, update_iters=iters, target_kl=0.02, kl_early_stop=False)
    t0 = run3(ac0, up0)
    assert up0.last_path == 'per-step' and up0._ug.get('graph') is not None
    # ---- the same three updates as a world of one rank over RCCL
    _env(port, 'replicated')
    from omnisafe_amd import distributed as dist

    dist.init_from_env('cuda:0')
    assert torch.distributed.get_backend() == 'nccl' and dist.collectives_active() and dist.graph_capturable()
    ac1 = make_ac(60, 2)
    ac1.params.copy_(p_init)
    up1 = PPOUpdater(ac1, batch_size=B, update_iters=iters, target_kl=0.02, kl_early_stop=False)
    t1 = run3(ac1, up1)
    assert up1.last_path == 'dp-large-batch-graph', up1.last_path
    err = float((ac1.params - ac0.params).abs().max())
    moved = float((ac0.params - p_init).abs().max())
    assert moved > 1e-3 and err <= 2e-7, (err, moved)
    assert ac1.adam_step.tolist() == ac0.adam_step.tolist() == [3 * iters * (M // B)] * 3
    # eager launches of the same path (what gloo gets): same bits as the captured pass
    os.environ['OSA_UPDATE_GRAPH'] = '0'
    ac2 = make_ac(60, 2)
    ac2.params.copy_(p_init)
    up2 = PPOUpdater(ac2, batch_size=B, update_iters=iters, target_kl=0.02, kl_early_stop=False)
    t2 = run3(ac2, up2)
    os.environ.pop('OSA_UPDATE_GRAPH')
    assert up2.last_path == 'dp-large-batch'
    assert torch.equal(ac2.params, ac1.params), float((ac2.params - ac1.params).abs().max())
    res = {'M': M, 'B': B, 'passes': iters, 'single_process_graph_us_per_step': round(t0, 2),
           'dp_large_batch_graph_us_per_step_world1_rccl': round(t1, 2),
           'dp_large_batch_eager_us_per_step_world1_rccl': round(t2, 2),
           'exchange_us_per_step_world1': round(t1 - t0, 2), 'max_abs_param_diff_vs_single_process': err,
           'bit_identical_to_single_process': bool(torch.equal(ac1.params, ac0.params)),
           'note': 'incl. the KL pass and the shuffles of an update (same on both sides); exchange = the difference: '
                   'grads-only finalize + RCCL all-reduce of 3 x 8448 floats + osa_adam_apply, no wire time at world 1'}
    torch.distributed.destroy_process_group()
    with open(out_path, 'w') as f:
        json.dump(res, f, indent=1)


def test_large_batch_dp_pass_is_one_graph_with_its_rccl_allreduces(tmp_path):
    out = str(tmp_path / 'lb.json')
    mp.spawn(_large_batch_graph_worker, args=(_free_port(), out), nprocs=1, join=True)
    res = json.load(open(out))
    print('large-batch DP step (world 1 over RCCL):', json.dumps(res))
    dst = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(dst, exist_ok=True)
    json.dump(res, open(os.path.join(dst, 'r4_dp_large_batch_world1.json'), 'w'), indent=1)


def _multi_device_worker(rank, world, port, algo, dp_mode, tmpdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), OSA_DP_MODE=dp_mode, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('OSA_DIST_BACKEND', 'OSA_SINGLE_DEVICE_RANKS', 'OSA_DIST_FORCE_COLLECTIVES'):
        os.environ.pop(k, None)
    sys.path.insert(0, ROOT)
    import omnisafe_amd
    from omnisafe_amd import distributed as dist

    cfg = {'seed': 4, 'train_cfgs': {'device': f'cuda:{rank}', 'total_steps': 2 * world * 128 * 16, 'vector_env_nums': 128},
           'algo_cfgs': {'steps_per_epoch': world * 128 * 16, 'update_iters': 2},
           'logger_cfgs': {'log_dir': os.path.join(tmpdir, f'r{rank}'), 'verbose': False},
           'env_cfgs': {'horizon': 8, 'cost_p': 0.3}}
    agent = omnisafe_amd.Agent(algo, 'SynthPointGoal1-v0', custom_cfgs=cfg)
    assert torch.distributed.get_backend() == 'nccl' and dist.world_size() == world
    p = agent.agent._actor_critic.params
    assert p.device.index == rank
    p0 = p.clone()
    ep_ret, ep_cost, ep_len = agent.learn()
    assert bool(torch.isfinite(p).all()) and not torch.equal(p, p0)
    assert ep_len == 8.0 and 1.0 < ep_cost < 4.0
    lo, hi = p.clone(), p.clone()  # every replica took the same optimiser steps
    torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
    torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
    assert torch.equal(lo, hi), 'replicas diverged'
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4, 8])
@pytest.mark.parametrize('algo,dp_mode', [('PPOLag', 'replicated'), ('PPOLag', 'allreduce'), ('TRPOLag', 'replicated'),
                                          ('CPO', 'allreduce')])
def test_agents_on_distinct_devices_over_rccl(tmp_path, world, algo, dp_mode):
    """Whole `Agent.learn()` runs with one rank per GPU over RCCL (skipped unless the box has `world` GPUs): rollouts
    sharded over the ranks' envs (policy_gradient.py:73-77), Lagrange / advantage statistics and gradients exchanged
    over the real xGMI collectives, replicas bit-identical afterwards."""
    if torch.cuda.device_count() < world:
        pytest.skip(f'needs {world} GPUs, this box has {torch.cuda.device_count()}')
    mp.spawn(_multi_device_worker, args=(world, _free_port(), algo, dp_mode, str(tmp_path)), nprocs=world, join=True)


def test_bench_launches_its_own_ranks(tmp_path):
    """The driver's scaling command is `python bench.py --gpus N ...` WITHOUT torchrun's environment: bench.py must start
    the N ranks itself (as the reference's fork(), omnisafe/utils/distributed.py:121-137) and rank 0 must print ONE
    JSON line for the whole job.  With N GPUs on the box: over RCCL.  On the 1-GPU test box: the documented hook
    (all ranks on cuda:0 over gloo) -- a check of the launch path and the line, not a measurement."""
    import subprocess

    n = 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'OSA_DIST_BACKEND', 'OSA_DP_MODE',
              'OSA_DIST_FORCE_COLLECTIVES'):
        env.pop(k, None)
    real = torch.cuda.device_count() >= n
    if not real:
        env['OSA_SINGLE_DEVICE_RANKS'] = '1'
    else:
        env.pop('OSA_SINGLE_DEVICE_RANKS', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '1', '--warmup', '2', '--envs',
           '256', '--update-iters', '2', '--allreduce-steps', '1', '--variant-batch', '2048', '--n1-value', '50000']
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == n and out['config']['parallelism'] == f'dp{n}' and out['config']['batch_size'] == 64
    assert out['config']['envs_per_gpu'] == 256 and out['config']['update_iters'] == 2
    assert out['config']['dp_mode'] == 'replicated' and out['config']['update_path'].startswith('replicated')
    assert out['rccl_ranks'] == (n if real else 0) and out['dist_backend'] == ('nccl' if real else 'gloo')
    assert out['value'] > 0 and 'efficiency_vs_n1' in out
    assert out['allreduce_mode']['update_path'] == ('allreduce-graph' if real else 'per-step')
    assert out['allreduce_mode']['value'] > 0
    assert out['throughput_variant']['update_path'].startswith('dp-large-batch')
    # (round 6) the one-shot peer exchange leg: ran, through osa_ppo_p2p_pass, and no guarded leg swallowed an exception
    assert out['p2p_mode'].get('update_path') == 'p2p' and out['p2p_mode']['value'] > 0, out['p2p_mode']
    assert not [k for k in ('allreduce_mode', 'p2p_mode', 'throughput_variant') if 'error' in out[k]]
    # refused, loudly, when the box has fewer GPUs and the hook is not set
    if not real:
        env.pop('OSA_SINGLE_DEVICE_RANKS')
        q = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
        assert q.returncode == 2 and 'OSA_SINGLE_DEVICE_RANKS' in q.stderr
