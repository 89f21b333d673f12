"""Debug tool (GPU): per-phase s_memtime deltas of one osa_ppo_minibatch launch (B=64)."""
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
OD, AD = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (60, 2)
ac = ConstraintActorCritic(Box(-np.inf, np.inf, (OD,)), Box(-1, 1, (AD,)), mc, 4, device=dev)
M = 65536
data = {'obs': torch.randn(M, OD, device=dev), 'act': torch.randn(M, AD, device=dev), 'logp': torch.randn(M, device=dev) - 2,
        'target_value_r': torch.randn(M, device=dev), 'target_value_c': torch.randn(M, device=dev),
        'adv_r': torch.randn(M, device=dev), 'adv_c': torch.randn(M, device=dev)}
up = PPOUpdater(ac, batch_size=64, update_iters=1, target_kl=0.02, kl_early_stop=False)
up.hp.lr_actor = up.hp.lr_critic = 3e-4
lam = torch.zeros(1, device=dev); stats = torch.zeros(16, device=dev)
perm = torch.randperm(M, device=dev)
for k in range(50):
    up.minibatch(data, perm[k * 64:(k + 1) * 64], 64, lam, stats)
dbg = torch.zeros(48, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.osa_debug_set_clock_buffer(dbg.data_ptr())
acc = np.zeros((3, 12))
R = 20
for k in range(R):
    up.minibatch(data, perm[(60 + k) * 64:(61 + k) * 64], 64, lam, stats)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(3, 16)
    acc += np.diff(d[:, :13], axis=1)
lib.osa_debug_set_clock_buffer(None)
names = ['fwd', 'loss', 'bwd', 'lds-transpose+sync', 'dW2+dW1', 'dW3', 'bias', 'loss-reduce', 'fin:loop1', 'fin:reduce', 'fin:clip', 'fin:adam(pow)+loop']
acc /= R
print('phase (s_memtime ticks @100MHz?)  actor  V_r  V_c')
for i, n in enumerate(names):
    print(f'{n:24s}', *[f'{v:9.0f}' for v in acc[:, i]])
print('total', acc.sum(1))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for k in range(1000):
    up.minibatch(data, perm[k * 64:(k + 1) * 64], 64, lam, stats)
e1.record(); torch.cuda.synchronize()
print('us per launch', e0.elapsed_time(e1))

if OD > 96: sys.exit(0)
# ---- persistent pass kernel
dbg2 = torch.zeros(48, dtype=torch.int64, device=dev)
lib.osa_debug_set_pass_clock_buffer(dbg2.data_ptr())
st = torch.zeros(1024, 16, device=dev)
up.run_pass(data, perm, lam, st); torch.cuda.synchronize()
e0.record(); up.run_pass(data, perm, lam, st); e1.record(); torch.cuda.synchronize()
lib.osa_debug_set_pass_clock_buffer(None)
d = dbg2.cpu().numpy().reshape(3, 16)[:, :10] / 1024.0
names2 = ['prefetch-issue', 'fwd', 'loss', 'bwd', 'transpose+barA', 'dW', 'bias+norms', 'barB', 'adam', 'stats+barC']
print('PASS kernel: cycles per minibatch   actor  V_r  V_c')
for i, n in enumerate(names2):
    print(f'{n:24s}', *[f'{v:9.0f}' for v in d[:, i]])
print('total', d.sum(1), 'us per minibatch (event)', e0.elapsed_time(e1) * 1e3 / 1024)

