"""GPU tool: time one pass of the replicated-data DP update for W virtual ranks on ONE GPU.  Each real rank of
a W-GPU run executes exactly this work, so this predicts the per-rank update time of the multi-GPU bench."""
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
M, B = 65536, 64
for W in (1, 2, 4, 8):
    ac = ConstraintActorCritic(Box(-np.inf, np.inf, (60,)), Box(-1, 1, (2,)), mc, 4, device=dev)
    data = {'obs': torch.randn(W * M, 60, device=dev), 'act': torch.randn(W * M, 2, device=dev), 'logp': torch.randn(W * M, device=dev) - 2,
            'target_value_r': torch.randn(W * M, device=dev), 'target_value_c': torch.randn(W * M, device=dev),
            'adv_r': torch.randn(W * M, device=dev), 'adv_c': torch.randn(W * M, device=dev)}
    lam = torch.zeros(1, device=dev); st = torch.zeros(1024, 16, device=dev)
    # cooperative pass with one XCC per network (L2 hand-offs) / spread over the XCCs with an uncached exchange
    # buffer; then the two-launches-per-step graph
    for coop, use_graph, xch in ((True, False, 'local'), (True, False, 'uncached'), (False, True, 'uncached')):
        os.environ['OSA_DP_XCH'] = xch
        up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False)
        up.hp.lr_actor = up.hp.lr_critic = 3e-4
        for _ in range(2):
            up.run_pass_replicated(data, M, W, lam, st, use_graph=use_graph, coop=coop)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); up.run_pass_replicated(data, M, W, lam, st, use_graph=use_graph, coop=coop); e1.record(); torch.cuda.synchronize()
        up.check_dp_sync()
        if coop:
            lib = _lib.load(); dbg = torch.zeros(48, dtype=torch.int64, device=dev)
            lib.osa_debug_set_pass_clock_buffer(dbg.data_ptr())
            up.run_pass_replicated(data, M, W, lam, st, coop=True); torch.cuda.synchronize()
            lib.osa_debug_set_pass_clock_buffer(None)
            d = dbg.cpu().numpy().reshape(3, 16)[:, :13] / 1024.0
            names = ['prefetch', 'fwd', 'loss', 'bwd', 'transpose+barA', 'dW', 'bias+norms', 'barB', 'adam', 'stats+barC', 'wait+acquire', 'reduce', 'publish+release']
            print('   cycles/step (actor, V_r, V_c): ' + '  '.join(f'{n}={d[0, i]:.0f}/{d[1, i]:.0f}/{d[2, i]:.0f}' for i, n in enumerate(names)))
        print(f'W={W} coop={coop} graph={use_graph} xch={xch if coop else "-"}: {e0.elapsed_time(e1):8.2f} ms per pass = {e0.elapsed_time(e1)*1e3/1024:6.2f} us per optimiser step', flush=True)
