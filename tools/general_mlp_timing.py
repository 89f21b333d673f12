"""GPU tool (round 4): the layer-wise path for general networks (csrc/general_mlp.hip) -- microseconds and TFLOP/s
per optimiser step (all three networks: forward + backward + clip + Adam) against hidden width and minibatch size,
inside a hipGraph as the updater runs it; under `rocprofv3 --kernel-trace --stats` the per-kernel split.

    python tools/general_mlp_timing.py [--out gpurun_out/r4_general_mlp_timing.json] [--shapes 1024x1024:16384 ...]
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
dev = 'cuda:0'
PEAK = 157.3


def weights(sizes):
    return sum(a * b for a, b in zip(sizes[:-1], sizes[1:]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='')
    ap.add_argument('--shapes', nargs='*', default=['1024x1024:64', '1024x1024:1024', '1024x1024:16384', '256x256:16384',
                                                    '512x512:4096', '64x64:16384'])
    ap.add_argument('--reps', type=int, default=20)
    args = ap.parse_args()
    d_o, d_a, M = 60, 2, 65536
    data = {'obs': torch.randn(M, d_o, device=dev), 'act': torch.randn(M, d_a, device=dev),
            'logp': torch.randn(M, device=dev) - 2, 'target_value_r': torch.randn(M, device=dev),
            'target_value_c': torch.randn(M, device=dev), 'adv_r': torch.randn(M, device=dev),
            'adv_c': torch.randn(M, device=dev)}
    lam = torch.zeros(1, device=dev)
    perm = torch.randperm(M, device=dev)
    os.environ['OSA_FORCE_GENERAL_MLP'] = '1'
    rows = []
    for spec in args.shapes:
        hs, B = spec.split(':')
        hid = [int(x) for x in hs.split('x')]
        B = int(B)
        mc = ns(actor=ns(hidden_sizes=hid, activation='tanh', lr=3e-4), critic=ns(hidden_sizes=hid, activation='tanh', lr=3e-4),
                weight_initialization_mode='kaiming_uniform', actor_type='gaussian_learning', linear_lr_decay=True)
        ac = ConstraintActorCritic(Box(-np.inf, np.inf, (d_o,)), Box(-1, 1, (d_a,)), mc, 4, device=dev)
        up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False)
        up.hp.lr_actor = up.hp.lr_critic = 3e-4
        row = torch.zeros(16, device=dev)
        idx = perm[:B].contiguous()
        for _ in range(2):
            up.minibatch(data, idx, B, lam, row)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        inner = 10 if B <= 1024 else 2
        with torch.cuda.graph(g):
            for _ in range(inner):
                up.minibatch(data, idx, B, lam, row)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (args.reps * inner)
        W = weights([d_o] + hid + [d_a]) + 2 * weights([d_o] + hid + [1])
        flops = 6 * W * B
        rec = {'hidden_sizes': hid, 'B': B, 'us_per_step': round(us, 2), 'GFLOP_per_step': round(flops / 1e9, 3),
               'TFLOPs': round(flops / us / 1e6, 2), 'frac_f32_mfma_peak': round(flops / us / 1e6 / PEAK, 4),
               'weight_bytes_3_nets_MB': round(4 * W / 1e6, 2)}
        rows.append(rec)
        print(json.dumps(rec), flush=True)
        del up, ac, g
    if args.out:
        json.dump({'device': torch.cuda.get_device_name(0), 'obs_act': [d_o, d_a], 'rows': rows}, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
