import os, sys, types, hashlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnisafe_amd.models import ConstraintActorCritic
from omnisafe_amd.spaces import Box
from omnisafe_amd.update import PPOUpdater
ns = types.SimpleNamespace
mc = ns(actor=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4), critic=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4),
        weight_initialization_mode='kaiming_uniform', actor_type='gaussian_learning', linear_lr_decay=True)
dev = 'cuda:0'
for (M, B) in ((65536, 16384), (10000, 10000), (2048, 2048), (6000, 6000), (40000, 40000)):
    torch.manual_seed(3)
    data = {'obs': torch.randn(M, 60, device=dev), 'act': torch.randn(M, 2, device=dev), 'logp': torch.randn(M, device=dev) - 2,
            'target_value_r': torch.randn(M, device=dev), 'target_value_c': torch.randn(M, device=dev),
            'adv_r': torch.randn(M, device=dev), 'adv_c': torch.randn(M, device=dev)}
    ac = ConstraintActorCritic(Box(-np.inf, np.inf, (60,)), Box(-1, 1, (2,)), mc, 4, device=dev)
    up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False)
    up.hp.lr_actor = up.hp.lr_critic = 3e-4
    up.hp.use_max_grad_norm = 0
    row = torch.zeros(16, device=dev); lam = torch.zeros(1, device=dev); idx = torch.randperm(M, device=dev)[:B].contiguous()
    up.minibatch(data, idx, B, lam, row)
    torch.cuda.synchronize()
    print(M, B, hashlib.sha256(ac.adam_m.cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha256(row[:5].cpu().numpy().tobytes()).hexdigest()[:8])
