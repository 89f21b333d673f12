"""GPU tool: average time per optimiser step of the cooperative data-parallel pass for W = 1, 2, 4, 8 virtual
ranks (5 timed passes each).  Run once per library build (OSA_LIB_PATH) for a same-box A/B."""
import os, sys, types
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
out = []
for W in (1, 2, 4, 8):
    torch.manual_seed(0)
    ac = ConstraintActorCritic(Box(-np.inf, np.inf, (60,)), Box(-1, 1, (2,)), mc, 4, device=dev)
    data = {'obs': torch.randn(W * M, 60, device=dev), 'act': torch.randn(W * M, 2, device=dev), 'logp': torch.randn(W * M, device=dev) - 2,
            'target_value_r': torch.randn(W * M, device=dev), 'target_value_c': torch.randn(W * M, device=dev),
            'adv_r': torch.randn(W * M, device=dev), 'adv_c': torch.randn(W * M, device=dev)}
    up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False)
    up.hp.lr_actor = up.hp.lr_critic = 3e-4
    lam = torch.zeros(1, device=dev); st = torch.zeros(1024, 16, device=dev)
    for _ in range(2):
        up.run_pass_replicated(data, M, W, lam, st, coop=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); up.run_pass_replicated(data, M, W, lam, st, coop=True); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 1024)
    up.check_dp_sync()
    out.append(f'W={W}: {np.mean(ts):.2f} (min {min(ts):.2f})')
print(os.environ.get('OSA_LIB_PATH', 'tree'), ' us/step  ', '  '.join(out), '  checksum %.6f' % float(ac.params.double().abs().sum()))
