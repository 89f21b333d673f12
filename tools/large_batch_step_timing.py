"""GPU tool: the large-batch optimiser step (partial gradients on the pass kernel's machinery + slab reduce / clip /
Adam) against the minibatch size: the slope is the cost of a 64-row chunk per workgroup, the intercept what a step
costs before any row is touched.

    python tools/large_batch_step_timing.py [--out gpurun_out/r3_large_batch_step.json]
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnisafe_amd.models import ConstraintActorCritic  # noqa: E402
from omnisafe_amd.spaces import Box  # noqa: E402
from omnisafe_amd.update import PPOUpdater  # noqa: E402

ns = types.SimpleNamespace
mc = ns(actor=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4),
        critic=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4),
        weight_initialization_mode='kaiming_uniform', actor_type='gaussian_learning', linear_lr_decay=True)
dev = 'cuda:0'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    d_o, d_a, M = 60, 2, 65536
    ld = 60
    data = {'obs': torch.randn(M, ld, device=dev), 'act': torch.randn(M, d_a, device=dev),
            'logp': torch.randn(M, device=dev) - 2, 'target_value_r': torch.randn(M, device=dev),
            'target_value_c': torch.randn(M, device=dev), 'adv_r': torch.randn(M, device=dev),
            'adv_c': torch.randn(M, device=dev)}
    lam = torch.zeros(1, device=dev)
    perm = torch.randperm(M, device=dev)
    rows = []
    for B in (2048, 4096, 8192, 16384):
        ac = ConstraintActorCritic(Box(-np.inf, np.inf, (d_o,)), Box(-1, 1, (d_a,)), mc, 4, device=dev)
        up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False)
        up.hp.lr_actor = up.hp.lr_critic = 3e-4
        row = torch.zeros(16, device=dev)
        idx = perm[:B].contiguous()
        g = torch.cuda.CUDAGraph()
        for _ in range(3):
            up.minibatch(data, idx, B, lam, row)
        torch.cuda.synchronize()
        reps = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.graph(g):
            for _ in range(20):
                up.minibatch(data, idx, B, lam, row)
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps // 20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        chunks = B // 64
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        wgs = min(cus, 3 * chunks)  # balanced partial gradients: the 3 x chunks tasks shared by one workgroup per CU
        rec = {'B': B, 'us_per_step_in_graph': round(us, 2), 'workgroups': wgs,
               'chunk_tasks_per_workgroup': round(3 * chunks / wgs, 2)}
        rows.append(rec)
        print(rows[-1], flush=True)
        del up, ac, g
    out = {'device': torch.cuda.get_device_name(0), 'shape': '60/2', 'rows': rows}
    if args.out:
        json.dump(out, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
