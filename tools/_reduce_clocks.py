import os, sys, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnisafe_amd import _lib
from omnisafe_amd.models import ConstraintActorCritic
from omnisafe_amd.spaces import Box
from omnisafe_amd.update import PPOUpdater
ns = types.SimpleNamespace
mc = ns(actor=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4), critic=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4),
        weight_initialization_mode='kaiming_uniform', actor_type='gaussian_learning', linear_lr_decay=True)
dev = 'cuda:0'; M = 65536; B = 16384
lib = _lib.load(require_gpu=True)
data = {'obs': torch.randn(M, 60, device=dev), 'act': torch.randn(M, 2, device=dev), 'logp': torch.randn(M, device=dev) - 2,
        'target_value_r': torch.randn(M, device=dev), 'target_value_c': torch.randn(M, device=dev),
        'adv_r': torch.randn(M, device=dev), 'adv_c': torch.randn(M, device=dev)}
ac = ConstraintActorCritic(Box(-np.inf, np.inf, (60,)), Box(-1, 1, (2,)), mc, 4, device=dev)
up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False)
up.hp.lr_actor = up.hp.lr_critic = 3e-4
row = torch.zeros(16, device=dev); lam = torch.zeros(1, device=dev); idx = torch.randperm(M, device=dev)[:B].contiguous()
for _ in range(5): up.minibatch(data, idx, B, lam, row)
dbg = torch.zeros(64, dtype=torch.int64, device=dev)
lib.osa_debug_set_clock_buffer(dbg.data_ptr())
acc = np.zeros((3, 7))
n = 20
for _ in range(n):
    up.minibatch(data, idx, B, lam, row); torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(4, 16)[:3, :7].astype(np.float64)
    acc += d - d[:, :1]
lib.osa_debug_set_clock_buffer(None)
print('reduce kernel, workgroup 0 of each network: clock64 ticks since entry at marks', (acc / n).round(0).tolist())
print('(1 slabs summed, 2 moments + pow requested, 3 block sums done / partials published, 4 barrier passed, 5 totals known, 6 Adam applied)')
