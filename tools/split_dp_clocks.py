"""GPU tool: osa_ppo_split_dp_pass at 376 / 17 for W virtual ranks -- microseconds per optimiser step and, with a
library built with -DOSA_SPLIT_CLOCKS (tools/build_variant_lib.sh sclocks wide_split_kernel.hip -DOSA_SPLIT_CLOCKS;
OSA_LIB_PATH=...), the phase clocks of rank 0's leader and helper 0 including the three phases of the cross-rank
average (publish + release | arrive + wait + acquire | slab reads)."""
import os, sys, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnisafe_amd.models import ConstraintActorCritic
from omnisafe_amd.spaces import Box
from omnisafe_amd.update import PPOUpdater
ns = types.SimpleNamespace
mc = ns(actor=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4), critic=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4),
        weight_initialization_mode='kaiming_uniform', actor_type='gaussian_learning', linear_lr_decay=True)
dev, M, B, d_o, d_a = 'cuda:0', 16384, 64, 376, 17
clocks = 'clocks' in os.environ.get('OSA_LIB_PATH', '')
for W in [int(x) for x in (sys.argv[1:] or ['1', '2', '4', '8'])]:
    for mode in ('place', 'spread'):
        os.environ['OSA_WIDE_DP'] = mode
        ac = ConstraintActorCritic(Box(-np.inf, np.inf, (d_o,)), Box(-1, 1, (d_a,)), mc, 4, device=dev)
        data = {'obs': torch.randn(W * M, d_o, device=dev), 'act': torch.randn(W * M, d_a, device=dev),
                'logp': torch.randn(W * M, device=dev) - 20, 'target_value_r': torch.randn(W * M, device=dev),
                'target_value_c': torch.randn(W * M, device=dev), 'adv_r': torch.randn(W * M, device=dev),
                'adv_c': torch.randn(W * M, device=dev)}
        lam = torch.zeros(1, device=dev); st = torch.zeros(M // B, 16, device=dev)
        up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False)
        up.hp.lr_actor = up.hp.lr_critic = 3e-4
        up._repl_wide = True
        for _ in range(2):
            up.run_pass_replicated(data, M, W, lam, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); up.run_pass_replicated(data, M, W, lam, st); e1.record(); torch.cuda.synchronize()
        up.check_wide_dp_sync()
        print(f'W={W} {mode}: {e0.elapsed_time(e1) * 1e3 / (M // B):7.2f} us per optimiser step', flush=True)
        if clocks:
            s = st.cpu().numpy()
            ln = ['wait partials', 'sum+forward', 'loss+backward+dz stores', 'publish dz1', 'dW2/dW3/bias/norm share', 'barrier+put', 'wait norm shares', 'DP average+Adam+barrier']
            hn = ['stage x+partial', 'publish', 'wait dz1', 'dW1', 'norm share', 'norm all-gather', 'DP average+Adam+barrier']
            dn = ['publish+release', 'arrive+wait+acquire', 'slab reads+sum']
            for net in range(3):
                print(f'   net {net} leader : ' + '  '.join(f'{n}={v:.0f}' for n, v in zip(ln, s[net, :8])) + f'  total={s[net, :8].sum():.0f}')
                print(f'   net {net} leader DP average: ' + '  '.join(f'{n}={v:.0f}' for n, v in zip(dn, s[3 + net, :3])))
                print(f'   net {net} helper0: ' + '  '.join(f'{n}={v:.0f}' for n, v in zip(hn, s[net, 8:15])) + f'  total={s[net, 8:15].sum():.0f}')
                print(f'   net {net} helper0 DP average: ' + '  '.join(f'{n}={v:.0f}' for n, v in zip(dn, s[3 + net, 8:11])))
        del up, ac, data
