"""GPU tool (round 6): microseconds per optimiser step of the ONE-SHOT PEER EXCHANGE (osa_ppo_p2p_pass, dp_mode 'p2p')
with W REAL ranks -- W processes, each with its own HIP context, exchange buffers mapped into each other through
hipIpcGetMemHandle / hipIpcOpenMemHandle.  On the 1-GPU box all W processes share device 0 (each rank's three
workgroups on their own compute units); with `--distinct-devices` rank r runs on device r (the driver's 8-GPU node).

    python tools/p2p_timing.py [--worlds 1 2 4 8] [--rows 16384] [--out gpurun_out/r6_p2p_timing.json]

Per W and shape: best-of-`reps` wall time of one pass (HIP events on rank 0, all ranks started together behind a
barrier) / steps of the pass, next to the single-GPU persistent pass of the same shape in the same process; the replicas'
parameters are compared bit for bit after the timed passes (all-reduce MIN / MAX over gloo).
"""
import argparse
import json
import os
import socket
import sys
import types

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [('config2 PPOLag 60/2', 60, 2, 64, 7), ('config5 TRPOLag critics 27/8', 27, 8, 128, 6),
          ('config3 CPO critics 72/2', 72, 2, 128, 6)]


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def worker(rank, world, port, rows, reps, distinct, out_path):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY='0', OSA_DIST_FORCE_COLLECTIVES='1')
    if not distinct:
        os.environ.update(OSA_DIST_BACKEND='gloo', OSA_SINGLE_DEVICE_RANKS='1')
    sys.path.insert(0, ROOT)
    import torch.distributed as tdist

    from omnisafe_amd import distributed as dist
    from omnisafe_amd.models import ConstraintActorCritic
    from omnisafe_amd.spaces import Box
    from omnisafe_amd.update import PPOUpdater

    dev = f'cuda:{rank}' if distinct else 'cuda:0'
    torch.cuda.set_device(dev)
    dist.init_from_env(dev)
    ns = types.SimpleNamespace
    mc = ns(actor=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4),
            critic=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4),
            weight_initialization_mode='kaiming_uniform', actor_type='gaussian_learning', linear_lr_decay=True)
    M = rows
    table = []
    for name, d_o, d_a, B, mask in SHAPES:
        nmb = (M + B - 1) // B
        torch.manual_seed(5)  # the same initial parameters on every rank ...
        ac = ConstraintActorCritic(Box(-np.inf, np.inf, (d_o,)), Box(-1, 1, (d_a,)), mc, 4, device=dev)
        torch.manual_seed(100 + rank)  # ... its own rows
        ld = (d_o + 3) // 4 * 4
        data = {'obs': torch.randn(M, ld, device=dev)[:, :d_o], 'act': torch.randn(M, d_a, device=dev),
                'logp': torch.randn(M, device=dev) - 2, 'target_value_r': torch.randn(M, device=dev),
                'target_value_c': torch.randn(M, device=dev), 'adv_r': torch.randn(M, device=dev),
                'adv_c': torch.randn(M, device=dev)}
        lam = torch.zeros(1, device=dev)
        up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False,
                        update_actor=(mask & 1) == 1, dp_mode='p2p')
        up.hp.lr_actor = up.hp.lr_critic = 3e-4
        assert up._p2p_ok(data) and up._p2p_setup(), 'peer exchange not available'
        st = torch.zeros(nmb, 16, device=dev)
        perm = torch.randperm(M, device=dev)
        best = 1e9
        for it in range(2 + reps):
            tdist.barrier()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            up.run_pass_p2p(data, perm, lam, st)
            e1.record()
            torch.cuda.synchronize()
            if it >= 2:
                best = min(best, e0.elapsed_time(e1))
        up.check_p2p_sync()
        if os.environ.get('OSA_LIB_PATH', '').endswith('p2pclocks.so') and rank == 0:
            # per-phase cycles of one optimiser step (build: tools/build_variant_lib.sh p2pclocks p2p_pass_kernel.hip
            # -DOSA_PASS_CLOCKS): the exchange = 'bias+norms .. publish' (stores issued while the norm is reduced),
            # 'release+barrier', 'arrival wait+acquire', 'sum of the slabs'
            dbg = torch.zeros(48, dtype=torch.int64, device=dev)
            up.lib.osa_debug_set_pass_clock_buffer(dbg.data_ptr())
        tdist.barrier()
        if os.environ.get('OSA_LIB_PATH', '').endswith('p2pclocks.so'):
            up.run_pass_p2p(data, perm, lam, st)
            torch.cuda.synchronize()
            if rank == 0:
                up.lib.osa_debug_set_pass_clock_buffer(None)
                d = dbg.cpu().numpy().reshape(3, 16)[:, :13] / float(nmb)
                names = ['top(mask,sX)', 'fwd(+prefetch issue)', 'loss', 'bwd', 'barA wait', 'dW', 'bias+norms(+peer stores)',
                         'barB', 'adam', 'stats+barC', 'arrival wait+acquire', 'sum of the slabs', 'release+barrier']
                order = [0, 1, 2, 3, 4, 5, 6, 7, 12, 10, 11, 8, 9]
                print(f'{name} W={world}: cycles per optimiser step   actor   V_r   V_c', flush=True)
                for i in order:
                    print(f'  {names[i]:30s}', *[f'{v:9.1f}' for v in d[:, i]], flush=True)
                print(f'  {"total":30s}', *[f'{v:9.1f}' for v in d.sum(1)], flush=True)
        p = ac.params.clone()
        lo, hi = p.cpu().clone(), p.cpu().clone()
        if world > 1:
            tdist.all_reduce(lo, op=tdist.ReduceOp.MIN)
            tdist.all_reduce(hi, op=tdist.ReduceOp.MAX)
        same = bool(torch.equal(lo, hi))
        # the single-GPU persistent pass of the same shape on the same rows (no exchange), this process alone
        single = None
        tdist.barrier()
        if rank == 0:
            ac1 = ConstraintActorCritic(Box(-np.inf, np.inf, (d_o,)), Box(-1, 1, (d_a,)), mc, 4, device=dev)
            os.environ['OSA_DIST_FORCE_COLLECTIVES'] = '0'
            up1 = PPOUpdater(ac1, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False,
                             update_actor=(mask & 1) == 1)
            up1.hp.lr_actor = up1.hp.lr_critic = 3e-4
            up1._pass_fn = ('osa_ppo_pass_kernel', up1.lib.osa_ppo_pass)
            b1 = 1e9
            for it in range(2 + reps):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                up1.run_pass(data, perm, lam, st)
                e1.record()
                torch.cuda.synchronize()
                if it >= 2:
                    b1 = min(b1, e0.elapsed_time(e1))
            single = b1 * 1e3 / nmb
            os.environ['OSA_DIST_FORCE_COLLECTIVES'] = '1'
        tdist.barrier()
        if rank == 0:
            rec = {'shape': name, 'batch_size': B, 'rows_per_rank': M, 'steps_per_pass': nmb, 'world': world,
                   'us_per_step_p2p': round(best * 1e3 / nmb, 2), 'us_per_step_single_gpu_pass': round(single, 2),
                   'single_gpu_path': up1.last_path or 'persistent', 'efficiency_vs_single_gpu_pass': round(single / (best * 1e3 / nmb), 3),
                   'replicas_bit_identical': same, 'distinct_devices': bool(distinct)}
            print(json.dumps(rec), flush=True)
            table.append(rec)
        del up
    if rank == 0:
        with open(out_path, 'w') as f:
            json.dump(table, f)
    tdist.barrier()
    tdist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=16384, help='rows per rank and pass')
    ap.add_argument('--worlds', type=int, nargs='+', default=[1, 2, 4, 8])
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--distinct-devices', action='store_true')
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    import tempfile

    res = []
    for W in args.worlds:
        if args.distinct_devices and torch.cuda.device_count() < W:
            continue
        tmp = tempfile.mktemp(suffix='.json')
        mp.spawn(worker, args=(W, _free_port(), args.rows, args.reps, args.distinct_devices, tmp), nprocs=W, join=True)
        res += json.load(open(tmp))
        os.unlink(tmp)
    out = {'device': torch.cuda.get_device_name(0),
           'note': 'W real processes (own HIP contexts, hipIpc-mapped exchange buffers); '
                   + ('rank r on device r' if args.distinct_devices else 'all on device 0: 3 workgroups per rank'),
           'table': res}
    print(json.dumps(out))
    if args.out:
        with open(args.out, 'w') as f:
            json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
