"""GPU tool (round 5): the per-step all-reduce mode of the update (gradient kernel -> ONE flat RCCL all-reduce -> Adam,
policy_gradient.py:437-443 with 1 message for 19) at the YAML batch, as a world of one rank over the real `nccl` backend:
microseconds per optimiser step eager and as the captured hipGraph of a pass; under `rocprofv3 --kernel-trace --stats`
the per-kernel split of a step.

    python tools/allreduce_step_trace.py [--out gpurun_out/r5_allreduce_step.json]
"""
import argparse
import json
import os
import socket
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='')
    ap.add_argument('--rows', type=int, default=65536)
    args = ap.parse_args()
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                      OSA_DIST_FORCE_COLLECTIVES='1', OSA_DP_MODE='allreduce')
    from omnisafe_amd import distributed as dist
    from omnisafe_amd.update import PPOUpdater
    from test_mlp_gpu import make_ac

    dist.init_from_env('cuda:0')
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    M, B = args.rows, 64
    ac = make_ac(60, 2)
    data = {'obs': torch.randn(M, 60, device=dev), 'act': torch.randn(M, 2, device=dev),
            'logp': torch.randn(M, device=dev) * 0.1 - 2.8, 'target_value_r': torch.randn(M, device=dev),
            'target_value_c': torch.randn(M, device=dev), 'adv_r': torch.randn(M, device=dev),
            'adv_c': torch.randn(M, device=dev)}
    lam = torch.tensor([0.2], device=dev)
    res = {'backend': torch.distributed.get_backend(), 'M': M, 'B': B}
    for mode in ('eager', 'graph'):
        if mode == 'eager':
            os.environ['OSA_UPDATE_GRAPH'] = '0'
        up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False, dp_mode='allreduce')
        for _ in range(4):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = up.run(data, lam, actor_lr=3e-4, critic_lr=3e-4)
            b.record()
            torch.cuda.synchronize()
        os.environ.pop('OSA_UPDATE_GRAPH', None)
        res[f'{mode}_us_per_step'] = round(a.elapsed_time(b) * 1e3 / out['steps'], 2)
        res[f'{mode}_path'] = up.last_path
    print(json.dumps(res))
    if args.out:
        json.dump(res, open(args.out, 'w'), indent=1)
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
