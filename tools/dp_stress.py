"""GPU tool: race hunt for the cooperative data-parallel passes.  The same pass (same rows, same permutation, same
initial parameters / Adam state) is executed `--iters` times; every execution must reproduce the first one BIT FOR BIT
(the replicas sum in rank order: nothing in the arithmetic depends on timing) and leave the sticky words at 0.
Between executions other kernels of varying length are enqueued so that the launches meet different device states.

    python tools/dp_stress.py [--iters 300] [--out gpurun_out/r3_dp_stress.json]
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnisafe_amd import update as U  # noqa: E402
from omnisafe_amd.models import ConstraintActorCritic  # noqa: E402
from omnisafe_amd.spaces import Box  # noqa: E402
from omnisafe_amd.update import PPOUpdater  # noqa: E402

ns = types.SimpleNamespace
mc = ns(actor=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4),
        critic=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4),
        weight_initialization_mode='kaiming_uniform', actor_type='gaussian_learning', linear_lr_decay=True)
dev = 'cuda:0'
# (name, obs, act, B, W, M, wide mode); W = 0: the single-GPU persistent pass of that shape (PPOUpdater.run_pass)
CASES = [('single wide split local', 376, 17, 64, 0, 1024, 'local'), ('single wide split spread', 376, 17, 64, 0, 1024, 'spread'),
         ('single chunked 72/2 B128', 72, 2, 128, 0, 1024, ''), ('single plain 60/2', 60, 2, 64, 0, 1024, ''),
         ('wide place W8', 376, 17, 64, 8, 192, 'place'), ('wide spread W8', 376, 17, 64, 8, 192, 'spread'),
         ('wide place W4', 376, 17, 64, 4, 256, 'place'), ('wide place W3 128/6', 128, 6, 64, 3, 200, 'place'),
         ('narrow placed W8', 60, 2, 64, 8, 192, ''), ('chunked W8 27/8', 27, 8, 128, 8, 256, ''),
         ('chunked W3 72/2', 72, 2, 128, 3, 300, '')]


def run_case(case, iters):
    """One row of CASES executed `iters` times from the same initial state; returns the report row."""
    name, d_o, d_a, B, W, M, wide_mode = case
    single = W == 0
    W = max(W, 1)
    if wide_mode:
        os.environ['OSA_WIDE_SPLIT' if single else 'OSA_WIDE_DP'] = wide_mode
    U._PLACEMENT['local_ok'] = None
    torch.manual_seed(W * 1000 + M)
    ld = (d_o + 3) // 4 * 4
    data = {'obs': torch.randn(W * M, ld, device=dev)[:, :d_o], 'act': torch.randn(W * M, d_a, device=dev),
            'logp': torch.randn(W * M, device=dev) * 0.3 - 2, 'target_value_r': torch.randn(W * M, device=dev) * 3,
            'target_value_c': torch.randn(W * M, device=dev), 'adv_r': torch.randn(W * M, device=dev),
            'adv_c': torch.randn(W * M, device=dev)}
    perms = torch.stack([torch.randperm(M) for _ in range(W)]).to(dev)
    lam = torch.tensor([0.4], device=dev)
    nmb = (M + B - 1) // B
    ac = ConstraintActorCritic(Box(-np.inf, np.inf, (d_o,)), Box(-1, 1, (d_a,)), mc, 4, device=dev)
    up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False, entropy_coef=0.01,
                    max_grad_norm=1.5)
    up.hp.lr_actor, up.hp.lr_critic = 3e-4, 1e-3
    if d_o > 96 and not single:
        assert up._wide_dp_fits(W)
        up._repl_wide = True
    init = [t.clone() for t in (ac.params, ac.adam_m, ac.adam_v, ac.adam_step)]
    if single:  # one whole update first: it selects the pass kernel and allocates its exchange buffer
        up.run({k: v[:M] for k, v in data.items()}, lam, actor_lr=3e-4, critic_lr=1e-3)
        up.hp.lr_actor, up.hp.lr_critic = 3e-4, 1e-3
    stats = torch.zeros(nmb, 16, device=dev)
    first, bad, errs = None, 0, []
    junk = torch.randn(1 << 22, device=dev)
    for it in range(iters):
        for t, s in zip((ac.params, ac.adam_m, ac.adam_v, ac.adam_step), init):
            t.copy_(s)
        if it % 3 == 1:  # a kernel of varying length right in front of the cooperative launch
            junk[: (it * 7919) % junk.numel() + 1].mul_(1.0001)
        if it % 5 == 2:
            torch.cuda.synchronize()
        try:
            if single:
                up.run_pass({k: v[:M] for k, v in data.items()}, perms[0], lam, stats)
                up.check_split_sync()
                up.check_chunk_sync()
            else:
                up.run_pass_replicated(data, M, W, lam, stats, perms_all=perms, use_graph=False, coop=True)
                if d_o > 96:
                    up.check_wide_dp_sync()
                else:
                    up.check_dp_sync()
        except Exception as e:  # noqa: BLE001
            errs.append(f'iter {it}: {type(e).__name__}: {e}')
            break
        out = torch.cat([ac.params.flatten(), ac.adam_m.flatten(), ac.adam_v.flatten(), stats.flatten()]).clone()
        if first is None:
            first = out
        elif not torch.equal(first, out):
            bad += 1
            if len(errs) < 8:
                d = (first - out).abs()
                P = ac.params.numel()
                per = P // 3
                where = []
                for ai, an in enumerate(('params', 'adam_m', 'adam_v')):
                    for n in range(3):
                        seg = d[ai * P + n * per: ai * P + (n + 1) * per]
                        if float(seg.max()) > 0:
                            where.append(f'{an}[net {n}]: {int((seg > 0).sum())} words, max {float(seg.max()):.2e}')
                sd = d[3 * P:].reshape(nmb, 16)
                rows = [(int(r), [int(c) for c in torch.nonzero(sd[r] > 0).flatten().tolist()]) for r in range(nmb)
                        if float(sd[r].max()) > 0]
                errs.append(f'iter {it}: {int((d > 0).sum())} words differ, max {float(d.max()):.3e}; ' +
                            '; '.join(where) + f'; stats rows/cols that differ: {rows}')
    if single:
        path = str(up.last_path) + (' (local)' if d_o > 96 and getattr(up, '_split_local', False) else '')
    elif d_o > 96:
        path = 'wide_place' if up._dp.get('wide_place') else 'wide_spread'
    else:
        path = ('chunked' if up._dp.get('chunked') else 'placed') + ('/local' if up._dp.get('local') else '/spread')
    return {'case': name, 'iters': iters, 'path': path, 'mismatching_runs': bad, 'messages': errs}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=300)
    ap.add_argument('--out', default='')
    ap.add_argument('--only', default='', help='substring of the case name')
    args = ap.parse_args()
    report = []
    for case in CASES:
        if args.only and args.only not in case[0]:
            continue
        rec = run_case(case, args.iters)
        report.append(rec)
        print(rec, flush=True)
    if not args.only or 'gae' in args.only:
        from omnisafe_amd.buffer import VectorOnPolicyBuffer
        for variant, T, N in (('chained', 1024, 4096), ('tiled', 256, 4096), ('chained', 4096, 512), ('chained', 5000, 4)):
            buf = VectorOnPolicyBuffer(Box(-np.inf, np.inf, (4,)), Box(-1, 1, (2,)), size=T, gamma=0.99, lam=0.95,
                                       lam_c=0.95, advantage_estimator='gae', penalty_coefficient=0.0,
                                       standardized_adv_r=True, standardized_adv_c=True, num_envs=N, device=dev,
                                       gae_variant=variant)
            for k in ('reward', 'value_r', 'value_c'):
                buf.data[k].normal_()
            buf.data['cost'].copy_((torch.rand(T, N, device=dev) < 0.05).float())
            pe = torch.rand(T, N, device=dev) < 0.02
            pe[-1] = True
            buf.data['path_end'].copy_(pe.to(torch.uint8))
            buf.data['boot_r'].copy_(torch.where(pe, torch.randn(T, N, device=dev), 0.0))
            buf.data['boot_c'].copy_(torch.where(pe, torch.randn(T, N, device=dev), 0.0))
            junk = torch.randn(1 << 22, device=dev)
            first, bad = None, 0
            for it in range(min(args.iters, 500)):
                buf.ptr = T
                if it % 3 == 1:
                    junk[: (it * 7919) % junk.numel() + 1].mul_(1.0001)
                buf.compute_advantages()
                out = torch.cat([buf.data[k].flatten() for k in ('adv_r', 'adv_c', 'target_value_r', 'target_value_c')]).clone()
                if first is None:
                    first = out
                elif not torch.equal(first, out):
                    bad += 1
            rec = {'case': f'gae {variant} T={T} N={N}', 'iters': min(args.iters, 500), 'mismatching_runs': bad}
            report.append(rec)
            print(rec, flush=True)
    if args.out:
        json.dump({'device': torch.cuda.get_device_name(0), 'cases': report}, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
