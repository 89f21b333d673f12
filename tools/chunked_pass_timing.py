#!/usr/bin/env python
"""us per optimiser step of the persistent pass for batch_size > 64 (the trust-region family's critic updates,
batch 128): the minibatch's 64-row chunks on cooperating workgroups (osa_ppo_chunked_pass, default) vs one
workgroup per network walking through the chunks (OSA_CHUNKED_PASS=0) vs the per-step launches.

    python tools/chunked_pass_timing.py [M]
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
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
res = {}
for obs_dim, act_dim, B, critics_only in ((72, 2, 128, True), (27, 8, 128, True), (60, 2, 128, False), (60, 2, 256, False),
                                          (60, 2, 512, False), (60, 2, 1024, False)):
    torch.manual_seed(0)
    data = {'obs': torch.randn(M, obs_dim, device=DEV), 'act': torch.randn(M, act_dim, device=DEV),
            'logp': torch.randn(M, device=DEV) * 0.1 - 3.0, 'target_value_r': torch.randn(M, device=DEV),
            'target_value_c': torch.randn(M, device=DEV), 'adv_r': torch.randn(M, device=DEV),
            'adv_c': torch.randn(M, device=DEV)}
    lam = torch.tensor([0.2], device=DEV)
    for persistent, chunked in ((True, '1'), (True, '0'), (False, '1')):
        os.environ['OSA_CHUNKED_PASS'] = chunked
        ac = make_ac(obs_dim, act_dim)
        up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False, persistent=persistent,
                        update_actor=not critics_only)
        up.persistent_max_batch = max(up.persistent_max_batch, int(os.environ.get('OSA_PMB', '512')))
        perm = [torch.randperm(M)]
        up.run(data, lam, perms=perm, actor_lr=3e-4, critic_lr=3e-4)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = up.run(data, lam, perms=perm, actor_lr=3e-4, critic_lr=3e-4)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / out['steps']
        tag = f'{obs_dim}/{act_dim} B={B} {"critics" if critics_only else "all nets"} {up.last_path}'
        res[tag] = round(us, 2)
        print(f'{tag:60s} {us:8.2f} us per optimiser step ({out["steps"]} steps)', flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'r3_chunked_pass_timing.json'), 'w'), indent=1)
