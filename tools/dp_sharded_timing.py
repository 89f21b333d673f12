"""GPU tool: the cooperative replicated-data pass with the three networks' chains SHARDED over the ranks
(profiles/HISTORY.md §5, "network-sharded" variant): a real rank of a W-GPU job then keeps only W workgroups
resident (one network, W virtual ranks) instead of 3 W, and the updated networks are broadcast once per pass.
Times one pass for W virtual ranks with nets_mask = all three networks / one network."""
import os, sys, types, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnisafe_amd.models import ConstraintActorCritic
from omnisafe_amd.spaces import Box
from omnisafe_amd.update import PPOUpdater
ns = types.SimpleNamespace
mc = ns(actor=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4), critic=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4),
        weight_initialization_mode='kaiming_uniform', actor_type='gaussian_learning', linear_lr_decay=True)
dev = 'cuda:0'
M, B = 65536, 64
res = {}
for W in (1, 2, 4, 8):
    ac = ConstraintActorCritic(Box(-np.inf, np.inf, (60,)), Box(-1, 1, (2,)), mc, 4, device=dev)
    data = {'obs': torch.randn(W * M, 60, device=dev), 'act': torch.randn(W * M, 2, device=dev), 'logp': torch.randn(W * M, device=dev) - 2,
            'target_value_r': torch.randn(W * M, device=dev), 'target_value_c': torch.randn(W * M, device=dev),
            'adv_r': torch.randn(W * M, device=dev), 'adv_c': torch.randn(W * M, device=dev)}
    lam = torch.zeros(1, device=dev); st = torch.zeros(1024, 16, device=dev)
    for label, mask in (('all three networks', 7), ('actor only', 1), ('one critic only', 2)):
        up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False)
        up.hp.lr_actor = up.hp.lr_critic = 3e-4
        up._nets_mask = (lambda m: (lambda: m))(mask)
        for _ in range(2):
            up.run_pass_replicated(data, M, W, lam, st, coop=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            up.run_pass_replicated(data, M, W, lam, st, coop=True)
        e1.record(); torch.cuda.synchronize()
        up.check_dp_sync()
        us = e0.elapsed_time(e1) * 1e3 / 3 / 1024
        res[f'W={W} {label}'] = round(us, 2)
        print(f'W={W} {label:22s}: {us:6.2f} us per optimiser step', flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(res, open('gpurun_out/r2_dp_sharded_timing.json', 'w'), indent=1)
