"""GPU tool: per-phase s_memtime cycles of the persistent pass kernel (needs the clocks build:
bash tools/build_variant_lib.sh clocks ppo_pass_kernel.hip -DOSA_PASS_CLOCKS; OSA_LIB_PATH=omnisafe_amd/lib/libomnisafe_amd_clocks.so python tools/pass_phases.py)."""
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
dev = 'cuda:0'
ac = ConstraintActorCritic(Box(-np.inf, np.inf, (60,)), Box(-1, 1, (2,)), mc, 4, device=dev)
M = 65536
data = {'obs': torch.randn(M, 60, device=dev), 'act': torch.randn(M, 2, device=dev), 'logp': torch.randn(M, device=dev) - 2,
        'target_value_r': torch.randn(M, device=dev), 'target_value_c': torch.randn(M, device=dev),
        'adv_r': torch.randn(M, device=dev), 'adv_c': torch.randn(M, device=dev)}
up = PPOUpdater(ac, batch_size=64, update_iters=1, target_kl=0.02, kl_early_stop=False)
up.hp.lr_actor = up.hp.lr_critic = 3e-4
lam = torch.zeros(1, device=dev)
perm = torch.randperm(M, device=dev)
lib = _lib.load()
dbg = torch.zeros(48, dtype=torch.int64, device=dev)
st = torch.zeros(1024, 16, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
up.run_pass(data, perm, lam, st); torch.cuda.synchronize()
lib.osa_debug_set_pass_clock_buffer(dbg.data_ptr())
e0.record(); up.run_pass(data, perm, lam, st); e1.record(); torch.cuda.synchronize()
lib.osa_debug_set_pass_clock_buffer(None)
d = dbg.cpu().numpy().reshape(3, 16)[:, :10] / 1024.0
names = ['top(mask,sX)', 'fwd(+prefetch issue)', 'loss', 'bwd', 'barA wait', 'dW', 'bias+norms', 'barB', 'adam', 'stats+barC']
print('cycles per optimiser step          actor      V_r      V_c')
for i, n in enumerate(names):
    print(f'{n:28s}', *[f'{v:9.0f}' for v in d[:, i]])
print(f'{"total":28s}', *[f'{v:9.0f}' for v in d.sum(1)], ' us per step (event):', round(e0.elapsed_time(e1) * 1e3 / 1024, 3))
