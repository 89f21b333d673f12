#!/usr/bin/env python
"""us per optimiser step of the wide-observation persistent passes (osa_ppo_split_pass: first layer split over
cooperating CUs, the default; osa_ppo_wide_pass: one CU per network, OSA_WIDE_SPLIT=0) vs the per-step kernels,
BASELINE config 4 shapes (376 / 17, batch 64) and other widths; one pass of M rows each.

    python tools/wide_pass_timing.py [M]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')]
from test_mlp_gpu import make_ac  # noqa: E402

from omnisafe_amd.update import PPOUpdater  # noqa: E402

DEV = 'cuda:0'
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
res = {}
for obs_dim, act_dim in ((376, 17), (128, 6), (512, 32), (90, 17), (200, 8)):
    torch.manual_seed(0)
    data = {'obs': torch.randn(M, obs_dim, device=DEV), 'act': torch.randn(M, act_dim, device=DEV),
            'logp': torch.randn(M, device=DEV) * 0.1 - 20.0, 'target_value_r': torch.randn(M, device=DEV),
            'target_value_c': torch.randn(M, device=DEV), 'adv_r': torch.randn(M, device=DEV),
            'adv_c': torch.randn(M, device=DEV)}
    lam = torch.tensor([0.2], device=DEV)
    for persistent, split in ((True, 'local'), (True, 'spread'), (True, '0'), (False, '1')):
        os.environ['OSA_WIDE_SPLIT'] = split
        ac = make_ac(obs_dim, act_dim)
        up = PPOUpdater(ac, batch_size=64, update_iters=1, target_kl=0.02, kl_early_stop=False,
                        persistent=persistent)
        perm = [torch.randperm(M)]
        up.run(data, lam, perms=perm, actor_lr=3e-4, critic_lr=3e-4)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = up.run(data, lam, perms=perm, actor_lr=3e-4, critic_lr=3e-4)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / out['steps']
        if persistent and split in ('local', 'spread') and 'sclocks' in os.environ.get('OSA_LIB_PATH', ''):
            st = out['stats'].cpu().numpy()
            ln = ['wait partials', 'sum+forward', 'loss+backward+dz stores', 'publish dz1', 'dW2/dW3/bias/norm share', 'barrier+put', 'wait norm shares', 'Adam+barrier']
            hn = ['stage x+partial', 'publish', 'wait dz1', 'dW1', 'norm share', 'norm all-gather', 'Adam+barrier']
            for net in range(3):
                row = st[net]
                print(f'   net {net} leader cycles/step: ' + '  '.join(f'{n}={v:.0f}' for n, v in zip(ln, row[:8])) + f'  total={row[:8].sum():.0f}')
                print(f'   net {net} helper0 cycles/step: ' + '  '.join(f'{n}={v:.0f}' for n, v in zip(hn, row[8:15])) + f'  total={row[8:15].sum():.0f}')
        elif persistent and split == '0' and 'wclocks' in os.environ.get('OSA_LIB_PATH', ''):
            st = out['stats'].cpu().numpy()
            names = ['fwd L1', 'fwd rest+loss+bwd', 'dW2/dW3/bias', 'dW1', 'norm+barrier', 'Adam W1', 'Adam rest+barrier']
            for net in range(3):
                row = st[out['steps'] - 1 - net, :7]
                print(f'   net {net} cycles/step: ' + '  '.join(f'{n}={v:.0f}' for n, v in zip(names, row)) + f'  total={row.sum():.0f}')
        tag = up.last_path + (f' ({split})' if up.last_path.endswith('split') else '')
        res[f'{obs_dim}/{act_dim} {tag}'] = round(us, 2)
        print(f'{obs_dim}/{act_dim}: {tag:32s} {us:8.2f} us per optimiser step ({out["steps"]} steps)', flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'r3_wide_pass_timing.json'), 'w'), indent=1)
