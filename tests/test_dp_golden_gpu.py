"""GPU: every data-parallel update path pinned to the UNMODIFIED REFERENCE run with two ranks.

tests/golden/dp2_<tag>.npz (oracle/make_golden.py::gen_dp2_updates) holds what each rank of a 2-rank gloo run of the
reference (`train_cfgs.parallel = 2`: utils/distributed.py:83-104; seeds base_algo.py:40; step sharding
policy_gradient.py:73-77) fed into and got out of one `_update()`: the rank's globally standardised `buf.get()` batch,
its minibatch permutations, its EpCost window, and the post-update parameters (identical on both ranks).  Here two
real ranks share the test box's GPU and talk over gloo (RCCL refuses two ranks on one device) -- the production code
path of world_size > 1: Lagrange step on the cross-rank mean cost, then

  replicated            all-gathered rollout + cooperative persistent pass (osa_ppo_dp_pass_placed), 60 / 2
  replicated-wide-split osa_ppo_split_dp_pass, 376 / 17 (BASELINE config 4)
  replicated + chunked  osa_ppo_dp_chunked_pass for TRPOLag's / CPO's batch-128 critics, 27 / 8 and 72 / 2 (BASELINE
                        configs 5 and 3), behind the
                        FVP / CG / line search whose products and losses are rank-averaged (natural_pg.py:112,
                        trpo.py:114-118, 181-185)
  replicated-steps      two launches per step from a hipGraph
  allreduce             gradient kernel -> ONE flat all-reduce -> Adam per optimiser step
  p2p                   the single-GPU persistent pass per rank, clipped gradients exchanged by one-shot peer writes into
                        hipIpc-mapped uncached buffers (osa_ppo_p2p_pass): no collective on the step path
  dp-large-batch        B = 2048: partial gradients -> local clip -> flat all-reduce -> Adam (graph-captured with RCCL;
                        eager over gloo)
  general-*             the same exchanges around the layer-wise GEMM path for general network shapes

each driven with the reference's inputs and compared with the reference's post-update parameters at the
single-process tolerances of tests/test_config_shapes_gpu.py (first-order family atol 2e-6 after 64 chained Adam
steps; TRPOLag: accepted line-search index identical, actor atol 5e-5, critics atol 2e-5)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
LAG = {'lagrangian_multiplier_init': 0.5, 'cost_limit': 0.5}


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tag, dp_mode, want_path, want, tmpdir, real_devices=False):
    os.environ.update(OSA_DP_MODE=dp_mode, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY='0')
    if real_devices:  # one rank per GPU over `nccl` (= RCCL over xGMI): the production configuration
        os.environ.pop('OSA_DIST_BACKEND', None)
        os.environ.pop('OSA_SINGLE_DEVICE_RANKS', None)
    else:  # all ranks on the test box's one GPU: RCCL refuses duplicate devices, so gloo (staged through the host)
        os.environ.update(OSA_DIST_BACKEND='gloo', OSA_SINGLE_DEVICE_RANKS='1')
    DEV = f'cuda:{rank}' if real_devices else 'cuda:0'
    if want_path.startswith('general-'):  # the layer-wise path for general networks, on the YAML-default shapes
        os.environ['OSA_FORCE_GENERAL_MLP'] = '1'
    sys.path.insert(0, ROOT)
    import omnisafe_amd
    from omnisafe_amd import distributed as dist

    g = dict(np.load(os.path.join(GOLDEN, f'{tag}.npz'), allow_pickle=False))
    N, T, algo_name, env_id = int(g['N']), int(g['T']), str(g['algo']), str(g['env_id'])
    assert int(g['world']) == world
    trust_region = algo_name in ('TRPOLag', 'CPO')
    M = N * T
    bs = 128 if trust_region else (2048 if tag.endswith('largebatch') else 64)
    cfg = {'seed': 0, 'train_cfgs': {'device': DEV, 'total_steps': 4 * world * M, 'vector_env_nums': N},
           'algo_cfgs': {'steps_per_epoch': world * M, 'update_iters': 2, 'kl_early_stop': False, 'batch_size': bs},
           'logger_cfgs': {'log_dir': os.path.join(tmpdir, f'r{rank}'), 'verbose': False}}
    if algo_name == 'CPO':
        cfg['algo_cfgs']['cost_limit'] = 0.5  # (as the golden: the constraint is violated -> the recovery case)
    else:
        cfg['lagrange_cfgs'] = LAG
    algo = omnisafe_amd.Agent(algo_name, env_id, custom_cfgs=cfg).agent
    assert dist.world_size() == world and algo._steps_per_epoch == T and algo._seed == 1000 * rank
    if real_devices:
        assert torch.distributed.get_backend() == 'nccl' and str(algo._actor_critic.device) == DEV
    ac = algo._actor_critic
    for net in ('actor', 'reward_critic', 'cost_critic'):
        sd = {k[len('init/') + len(net) + 1:]: torch.from_numpy(v.copy()) for k, v in g.items()
              if k.startswith(f'init/{net}/')}
        getattr(ac, net).load_state_dict(sd)
    data = {k[len(f'r{rank}/data/'):]: torch.from_numpy(np.ascontiguousarray(v)).to(DEV)
            for k, v in g.items() if k.startswith(f'r{rank}/data/')}
    assert data['obs'].shape[0] == M
    algo._buf.get = lambda: dict(data)
    algo._logger.extend('Metrics/EpCost', [float(v) for v in g[f'r{rank}/ep_cost_window']])
    perms = [g[f'r{r}/perms'] for r in range(world)]
    up = algo._updater
    # the replicated passes run ALL ranks' minibatches on every GPU: every rank needs every rank's order
    repl = dp_mode.startswith('replicated') and bs <= 512
    if repl:
        algo._perms_override = [torch.from_numpy(np.stack([perms[r][i] for r in range(world)])) for i in range(2)]
    else:
        algo._perms_override = [torch.from_numpy(perms[rank][i].copy()) for i in range(2)]
    algo._update()
    torch.cuda.synchronize()
    assert up.last_path == want_path, (up.last_path, want_path)
    for k, v in (want or {}).items():
        assert up._dp.get(k) == v, (k, up._dp.get(k), v)
    if hasattr(algo, '_lagrange'):
        np.testing.assert_allclose(algo._lagrange.lagrangian_multiplier, float(g['lambda_after']), rtol=1e-6)

    def max_err(net):
        return max(float(np.abs(v.cpu().numpy() - g[f'post/{net}/{k}']).max())
                   for k, v in getattr(ac, net).state_dict().items())

    errs = {n: max_err(n) for n in ('actor', 'reward_critic', 'cost_critic')}
    moved = max(float(np.abs(g[f'post/actor/{k}'] - g[f'init/actor/{k}']).max()) for k in ac.actor.state_dict())
    if rank == 0:
        print(tag, dp_mode, up.last_path, f'max |param - {world}-rank reference|:', errs, 'actor moved', moved, flush=True)
    assert moved > 1e-3
    if trust_region:
        lg = lambda key: np.asarray(list(algo._logger._data[key]), np.float64)  # noqa: E731
        assert int(lg('Misc/AcceptanceStep')[-1]) == int(g['r0/log/Misc/AcceptanceStep'][-1])
        if algo_name == 'CPO':  # the same case of the two-constraint problem (cpo.py:291-330), the same multipliers
            assert int(lg('Misc/OptimCase')[-1]) == int(g['r0/log/Misc/OptimCase'][-1])
            for key in ('Misc/q', 'Misc/r', 'Misc/s', 'Misc/Nu_star', 'Misc/cost_gradient_norm'):
                np.testing.assert_allclose(lg(key)[-1], g['r0/log/' + key][-1], rtol=2e-2, err_msg=key)
        for key, rtol in (('Misc/Alpha', 1e-2), ('Misc/xHx', 1e-2), ('Misc/gradient_norm', 1e-3),
                          ('Misc/FinalStepNorm', 2e-2)):
            np.testing.assert_allclose(lg(key)[-1], g['r0/log/' + key][-1], rtol=rtol, err_msg=key)
        assert errs['actor'] < 5e-5 and max(errs['reward_critic'], errs['cost_critic']) < 2e-5, errs
    elif env_id == 'SynthHumanoid-v0':
        # 376 inputs: the first layer is a 376-term float32 sum in MFMA-tile order (the split wide pass: over
        # cooperating workgroups) where the reference's CPU sgemm has its own order (~1e-7 relative in every gradient); through Adam's first steps lr g / (|g| + eps) the handful of elements whose
        # early gradients lie within ~1e-8 of zero move by a visible fraction of lr (profiles/HISTORY.md §3.3; the single-GPU
        # test_wide_split_* asserts the same shape): all but a few of the 86 k parameters inside the single-process
        # tolerance, the stragglers inside a twentieth of ONE learning-rate step
        big = sum(int((np.abs(v.cpu().numpy() - g[f'post/{net}/{k}']) > 2e-6).sum())
                  for net in ('actor', 'reward_critic', 'cost_critic') for k, v in getattr(ac, net).state_dict().items())
        assert big <= 8 and max(errs.values()) < 1.5e-5, (big, errs)
    else:
        if world >= 8:
            # eight-term rank sums in MFMA / rank order v gloo's ring order, through 64 chained Adam steps: all but a
            # handful of the 22 k parameters inside the single-process tolerance, the rest within a hundredth of ONE
            # learning-rate step (measured: 1 element at 2.2e-6)
            big = sum(int((np.abs(v.cpu().numpy() - g[f'post/{net}/{k}']) > 2e-6).sum())
                      for net in ('actor', 'reward_critic', 'cost_critic') for k, v in getattr(ac, net).state_dict().items())
            assert big <= 4 and max(errs.values()) < 5e-6, (big, errs)
        else:
            assert max(errs.values()) < 2e-6, errs
        np.testing.assert_allclose(float(algo._logger._data['Train/KL'][-1]), g['r0/log/Train/KL'][-1], rtol=1e-2,
                                   atol=1e-7)
    # replicas identical
    p = ac.params
    lo, hi = p.clone(), p.clone()
    torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
    torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
    assert torch.equal(lo, hi), 'replicas diverged'
    torch.distributed.destroy_process_group()


CASES = [
    ('dp2_ppolag_point', 'replicated', 'replicated', {'chunked': False}),
    ('dp2_ppolag_point', 'replicated-steps', 'replicated', None),
    ('dp2_ppolag_point', 'allreduce', 'per-step', None),
    ('dp2_ppolag_humanoid', 'replicated', 'replicated-wide-split', None),
    ('dp2_ppolag_humanoid', 'allreduce', 'per-step', None),
    ('dp2_trpolag_ant', 'replicated', 'replicated', {'chunked': True}),
    ('dp2_trpolag_ant', 'allreduce', 'per-step', None),
    # BASELINE config 3: CPO, 72 / 2 -- reward and cost gradients, two CG solves, the recovery case, all on rank averages
    ('dp2_cpo_car', 'replicated', 'replicated', {'chunked': True}),
    ('dp2_cpo_car', 'allreduce', 'per-step', None),
    ('dp2_ppolag_point_largebatch', 'replicated', 'dp-large-batch', None),
    ('dp2_ppolag_point_largebatch', 'allreduce', 'dp-large-batch', None),
    # ONE-SHOT PEER EXCHANGE (round 6, osa_ppo_p2p_pass): every rank runs the single-GPU persistent pass on its own rows,
    # the clipped gradients travel by peer writes into hipIpc-mapped uncached buffers (here: two processes on one GPU)
    ('dp2_ppolag_point', 'p2p', 'p2p', None),
    ('dp2_trpolag_ant', 'p2p', 'p2p', None),     # batch-128 critic passes: the workgroup walks through two chunks
    ('dp2_cpo_car', 'p2p', 'p2p', None),
    ('dp2_ppolag_humanoid', 'p2p', 'per-step', None),  # 376-wide rows: outside the pass kernel -> per-step all-reduce
    # general networks (csrc/general_mlp.hip) under data parallelism: gradient GEMMs -> local clip -> flat all-reduce
    # -> osa_gmlp_adam_apply per step; FVP / line-search averages of the trust-region family
    ('dp2_ppolag_point', 'allreduce', 'general-per-step', None),
    ('dp2_trpolag_ant', 'allreduce', 'general-per-step', None),
    ('dp2_ppolag_point_largebatch', 'allreduce', 'general-dp-large-batch', None),
]


@pytest.mark.parametrize('tag,dp_mode,want_path,want', CASES)
def test_two_ranks_reproduce_the_two_rank_reference(tmp_path, tag, dp_mode, want_path, want):
    mp.spawn(_worker, args=(2, _free_port(), tag, dp_mode, want_path, want, str(tmp_path)), nprocs=2, join=True)


@pytest.mark.parametrize('tag,dp_mode,want_path,want', [
    ('dp4_ppolag_point', 'replicated', 'replicated', {'chunked': False}),
    ('dp4_ppolag_point', 'allreduce', 'per-step', None),
    ('dp4_trpolag_ant', 'replicated', 'replicated', {'chunked': True}),
    ('dp4_ppolag_point', 'p2p', 'p2p', None),
    ('dp4_trpolag_ant', 'p2p', 'p2p', None),
])
def test_four_ranks_reproduce_the_four_rank_reference(tmp_path, tag, dp_mode, want_path, want):
    """The same recordings from a FOUR-rank run of the unmodified reference (`oracle/make_golden.py dp4`): a sum of four
    rank gradients is no longer order-free -- the cooperative pass adds them in rank order, gloo's ring in its own -- so
    this pins the rank-ordered reduction, the 1 / W scaling and the rank indexing beyond the two-rank case, at the same
    tolerances."""
    mp.spawn(_worker, args=(4, _free_port(), tag, dp_mode, want_path, want, str(tmp_path)), nprocs=4, join=True)


EIGHT = [
    ('dp8_ppolag_point', 'replicated', 'replicated', None),
    ('dp8_ppolag_point', 'allreduce', 'per-step', None),
    # BASELINE configs 4, 5, 3 at the world size BASELINE.json quotes them on (`make_golden.py dp8 dp2_ppolag_humanoid@32
    # dp2_trpolag_ant dp2_cpo_car`): the wide split pass with 8 x 3 x 5 workgroups, the chunked passes with 8 x 2 peers
    # per network behind rank-averaged Fisher-vector products / line searches, CPO's recovery case on 8-rank averages
    ('dp8_ppolag_humanoid', 'replicated', 'replicated-wide-split', None),
    ('dp8_ppolag_humanoid', 'allreduce', 'per-step', None),
    ('dp8_trpolag_ant', 'replicated', 'replicated', {'chunked': True}),
    ('dp8_trpolag_ant', 'allreduce', 'per-step', None),
    ('dp8_cpo_car', 'replicated', 'replicated', {'chunked': True}),
    ('dp8_cpo_car', 'allreduce', 'per-step', None),
    # the peer exchange at the world size BASELINE.json quotes: 8 processes x 3 workgroups, every workgroup writes its
    # slab into 8 buffers and adds 8 slabs in rank order
    ('dp8_ppolag_point', 'p2p', 'p2p', None),
    ('dp8_trpolag_ant', 'p2p', 'p2p', None),
    ('dp8_cpo_car', 'p2p', 'p2p', None),
]


@pytest.mark.parametrize('tag,dp_mode,want_path,want', EIGHT)
def test_eight_ranks_reproduce_the_eight_rank_reference(tmp_path, tag, dp_mode, want_path, want):
    """BASELINE.json quotes its 8-GPU configs at world size 8: the unmodified reference run with EIGHT ranks
    (`oracle/make_golden.py dp8`: PPOLag 60 / 2 and 376 / 17, TRPOLag 27 / 8, CPO 72 / 2) against eight real ranks
    sharing the test box's GPU -- the cooperative passes with 24 .. 120 resident workgroups, and the per-step
    all-reduce path."""
    mp.spawn(_worker, args=(8, _free_port(), tag, dp_mode, want_path, want, str(tmp_path)), nprocs=8, join=True)



# ---- the same recordings on DISTINCT devices over `nccl` (= RCCL): runs wherever the box has the GPUs (the driver's
# 8-GPU node), skips on the 1-GPU test box.  Over RCCL the large-batch pass is a captured graph incl. its all-reduces.
REAL = [
    (2, 'dp2_ppolag_point', 'replicated', 'replicated', {'chunked': False}),
    (2, 'dp2_ppolag_point', 'allreduce', 'per-step', None),
    (2, 'dp2_ppolag_humanoid', 'replicated', 'replicated-wide-split', None),
    (2, 'dp2_trpolag_ant', 'replicated', 'replicated', {'chunked': True}),
    (2, 'dp2_cpo_car', 'allreduce', 'per-step', None),
    (2, 'dp2_ppolag_point_largebatch', 'allreduce', 'dp-large-batch-graph', None),
    (4, 'dp4_ppolag_point', 'replicated', 'replicated', {'chunked': False}),
    (4, 'dp4_ppolag_point', 'allreduce', 'per-step', None),
    (8, 'dp8_ppolag_point', 'replicated', 'replicated', None),
    (8, 'dp8_ppolag_point', 'allreduce', 'per-step', None),
    (8, 'dp8_ppolag_humanoid', 'replicated', 'replicated-wide-split', None),
    (8, 'dp8_trpolag_ant', 'replicated', 'replicated', {'chunked': True}),
    (8, 'dp8_cpo_car', 'allreduce', 'per-step', None),
    (2, 'dp2_ppolag_point', 'p2p', 'p2p', None),
    (4, 'dp4_ppolag_point', 'p2p', 'p2p', None),
    (8, 'dp8_ppolag_point', 'p2p', 'p2p', None),
    (8, 'dp8_trpolag_ant', 'p2p', 'p2p', None),
]


@pytest.mark.parametrize('world,tag,dp_mode,want_path,want', REAL)
def test_ranks_on_distinct_devices_over_rccl_reproduce_the_reference(tmp_path, world, tag, dp_mode, want_path, want):
    """One rank per GPU, `nccl` backend (omnisafe/utils/distributed.py:75-80,100): the reference's multi-rank
    recordings through the real RCCL collectives -- flat gradient all-reduce (ReduceOp.AVG), the rollout all-gather of
    the replicated mode, the fp64 statistics, the captured large-batch pass."""
    if torch.cuda.device_count() < world:
        pytest.skip(f'needs {world} GPUs, this box has {torch.cuda.device_count()}')
    mp.spawn(_worker, args=(world, _free_port(), tag, dp_mode, want_path, want, str(tmp_path), True), nprocs=world,
             join=True)
