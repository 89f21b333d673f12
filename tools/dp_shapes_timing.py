"""GPU tool: microseconds per optimiser step of the data-parallel update passes for W virtual ranks on ONE GPU, at the
shapes of the BASELINE configs (each real rank of a W-GPU job executes exactly this work per pass), next to the
single-GPU persistent pass of the same shape:

    config 2  PPOLag   60 / 2    B 64    osa_ppo_pass            | osa_ppo_dp_pass_placed
    config 4  PPOLag   376 / 17  B 64    osa_ppo_split_pass      | osa_ppo_split_dp_pass        (round 3)
    config 5  TRPOLag  27 / 8    B 128   osa_ppo_chunked_pass    | osa_ppo_dp_chunked_pass      (round 3)
    config 3  CPO      72 / 2    B 128   osa_ppo_chunked_pass    | osa_ppo_dp_chunked_pass      (round 3)

    python tools/dp_shapes_timing.py [--rows 16384] [--out profiles/r3_dp_shapes_timing.json]
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnisafe_amd.models import ConstraintActorCritic  # noqa: E402
from omnisafe_amd.spaces import Box  # noqa: E402
from omnisafe_amd.update import PPOUpdater  # noqa: E402

ns = types.SimpleNamespace
mc = ns(actor=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4),
        critic=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4),
        weight_initialization_mode='kaiming_uniform', actor_type='gaussian_learning', linear_lr_decay=True)
dev = 'cuda:0'
SHAPES = [('config2 PPOLag 60/2', 60, 2, 64, 7), ('config4 PPOLag 376/17', 376, 17, 64, 7),
          ('config5 TRPOLag critics 27/8', 27, 8, 128, 6), ('config3 CPO critics 72/2', 72, 2, 128, 6)]


def make_data(rows, d_o, d_a):
    ld = (d_o + 3) // 4 * 4
    return {'obs': torch.randn(rows, ld, device=dev)[:, :d_o], 'act': torch.randn(rows, d_a, device=dev),
            'logp': torch.randn(rows, device=dev) - 2, 'target_value_r': torch.randn(rows, device=dev),
            'target_value_c': torch.randn(rows, device=dev), 'adv_r': torch.randn(rows, device=dev),
            'adv_c': torch.randn(rows, device=dev)}


def timed(fn, reps=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=16384, help='rows per rank and pass (M)')
    ap.add_argument('--worlds', type=int, nargs='+', default=[1, 2, 4, 8])
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    M = args.rows
    table = []
    for name, d_o, d_a, B, mask in SHAPES:
        nmb = (M + B - 1) // B
        rec = {'shape': name, 'obs_dim': d_o, 'act_dim': d_a, 'batch_size': B, 'rows_per_rank': M,
               'steps_per_pass': nmb, 'us_per_step': {}}
        # ---- single-GPU reference point: the persistent pass a 1-GPU job runs
        ac = ConstraintActorCritic(Box(-np.inf, np.inf, (d_o,)), Box(-1, 1, (d_a,)), mc, 4, device=dev)
        data = make_data(M, d_o, d_a)
        lam = torch.zeros(1, device=dev)
        up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False,
                        update_actor=(mask & 1) == 1)
        up.run(data, lam, actor_lr=3e-4, critic_lr=3e-4)
        torch.cuda.synchronize()
        ms = timed(lambda: up.run(data, lam, actor_lr=3e-4, critic_lr=3e-4))
        rec['single_gpu'] = {'path': up.last_path, 'us_per_step_incl_kl_and_perm': round(ms * 1e3 / nmb, 2)}
        st = torch.zeros(nmb, 16, device=dev)
        perm = torch.randperm(M, device=dev)
        up.hp.lr_actor = up.hp.lr_critic = 3e-4
        if up._pass_fn is not None:
            ms = timed(lambda: up.run_pass(data, perm, lam, st))
            rec['single_gpu']['us_per_step'] = round(ms * 1e3 / nmb, 2)
        del up, ac
        # wide shape: owner groups placed / spread
        modes = ['place', 'spread'] if d_o > 96 else ['']
        for W, wide_mode in [(w, m) for w in args.worlds for m in modes]:
            if wide_mode in ('place', 'spread'):
                os.environ['OSA_WIDE_DP'] = wide_mode
            ac = ConstraintActorCritic(Box(-np.inf, np.inf, (d_o,)), Box(-1, 1, (d_a,)), mc, 4, device=dev)
            data_all = make_data(W * M, d_o, d_a)
            up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False,
                            update_actor=(mask & 1) == 1)
            up.hp.lr_actor = up.hp.lr_critic = 3e-4
            wide = d_o > 96
            if wide:
                if not up._wide_dp_fits(W):
                    rec['us_per_step'][str(W)] = None
                    continue
                up._repl_wide = True
            st = torch.zeros(nmb, 16, device=dev)
            ms = timed(lambda: up.run_pass_replicated(data_all, M, W, lam, st, use_graph=False, coop=True))
            if wide:
                up.check_wide_dp_sync()
                path = 'osa_ppo_split_dp_pass (' + ('owner groups on one XCC, L2 exchange' if up._dp.get('wide_place')
                                                     else 'rank-major, uncached exchange') + ')'
            else:
                up.check_dp_sync()
                path = ('osa_ppo_dp_chunked_pass' if up._dp.get('chunked') else 'osa_ppo_dp_pass_placed') + \
                       (' (one XCC per network)' if up._dp.get('local') else ' (spread, uncached exchange)')
            key = str(W) if wide_mode in ('', 'place') else f'{W} ({wide_mode})'
            rec['us_per_step'][key] = round(ms * 1e3 / nmb, 2)
            rec.setdefault('dp_path', {})[key] = path
            print(f'{name}: W={W} {path}: {ms * 1e3 / nmb:7.2f} us per optimiser step '
                  f'(single GPU {rec["single_gpu"].get("us_per_step")} us, {rec["single_gpu"]["path"]})', flush=True)
            del up, ac, data_all
        one = rec['single_gpu'].get('us_per_step')
        if one:
            rec['efficiency_vs_single_gpu_pass'] = {w: (round(one / v, 3) if v else None)
                                                    for w, v in rec['us_per_step'].items()}
        table.append(rec)
    out = {'device': torch.cuda.get_device_name(0), 'note': 'W virtual ranks on one GPU: the per-rank work of a W-GPU job',
           'table': table}
    print(json.dumps(out))
    if args.out:
        with open(args.out, 'w') as f:
            json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
