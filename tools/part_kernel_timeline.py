"""GPU tool (round 4): per-workgroup timeline of the balanced partial-gradient launch of the large-batch step
(osa_ppo_part_kernel) from its debug clocks: dispatch skew, per-workgroup duration against the number of tasks it
owns (the slope = cost of a 64-row chunk, the intercept = prologue + epilogue), the straddling workgroups.

    python tools/part_kernel_timeline.py [--out gpurun_out/r4_part_timeline.json]
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnisafe_amd import _lib  # noqa: E402
from omnisafe_amd.models import ConstraintActorCritic  # noqa: E402
from omnisafe_amd.spaces import Box  # noqa: E402
from omnisafe_amd.update import PPOUpdater  # noqa: E402

ns = types.SimpleNamespace
mc = ns(actor=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4),
        critic=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4),
        weight_initialization_mode='kaiming_uniform', actor_type='gaussian_learning', linear_lr_decay=True)
dev = 'cuda:0'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    lib = _lib.load()
    M = 65536
    data = {'obs': torch.randn(M, 60, device=dev), 'act': torch.randn(M, 2, device=dev),
            'logp': torch.randn(M, device=dev) - 2, 'target_value_r': torch.randn(M, device=dev),
            'target_value_c': torch.randn(M, device=dev), 'adv_r': torch.randn(M, device=dev),
            'adv_c': torch.randn(M, device=dev)}
    lam = torch.zeros(1, device=dev)
    perm = torch.randperm(M, device=dev)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    out = {'device': torch.cuda.get_device_name(0), 'cus': cus, 'rows': []}
    for B in (2048, 4096, 8192, 12288, 16384, 32768):
        ac = ConstraintActorCritic(Box(-np.inf, np.inf, (60,)), Box(-1, 1, (2,)), mc, 4, device=dev)
        up = PPOUpdater(ac, batch_size=B, update_iters=1, target_kl=0.02, kl_early_stop=False)
        up.hp.lr_actor = up.hp.lr_critic = 3e-4
        row = torch.zeros(16, device=dev)
        idx = perm[:B].contiguous()
        for _ in range(3):
            up.minibatch(data, idx, B, lam, row)
        ntasks = 3 * (B // 64)
        G = min(cus, ntasks)
        tpw = -(-ntasks // G)
        G = -(-ntasks // tpw)
        dbg = torch.zeros(16 * G, dtype=torch.int64, device=dev)
        lib.osa_debug_set_part_clock_buffer(dbg.data_ptr())
        recs = []
        for _ in range(5):
            up.minibatch(data, idx, B, lam, row)
            torch.cuda.synchronize()
            d = dbg.cpu().numpy().reshape(G, 16).astype(np.float64)
            t0 = d[:, 0].min()
            start_us, end_us = (d[:, 0] - t0) / 100.0, (d[:, 2] - t0) / 100.0
            dur_us = end_us - start_us
            cyc = d[:, 3] - d[:, 1]
            recs.append((start_us, end_us, dur_us, cyc, d))
        lib.osa_debug_set_part_clock_buffer(None)
        start_us, end_us, dur_us, cyc, d = recs[-1]
        # tasks per workgroup and whether it straddles two networks
        nchunk = B // 64
        tasks = np.array([min((w + 1) * tpw, ntasks) - w * tpw for w in range(G)])
        straddle = np.array([(w * tpw) // nchunk != (min((w + 1) * tpw, ntasks) - 1) // nchunk for w in range(G)])
        rec = {'B': B, 'workgroups': G, 'tasks_per_workgroup': tpw, 'last_start_us': round(float(start_us.max()), 2),
               'kernel_span_us': round(float(end_us.max()), 2),
               'duration_us_plain_median': round(float(np.median(dur_us[~straddle])), 2),
               'duration_us_plain_max': round(float(dur_us[~straddle].max()), 2),
               'duration_us_straddling': [round(float(x), 2) for x in dur_us[straddle]],
               'shader_GHz_median': round(float(np.median(cyc / (dur_us * 1e3))), 3),
               'cycles_median': int(np.median(cyc)),
               'end_us_percentiles_50_90_100': [round(float(np.percentile(end_us, q)), 2) for q in (50, 90, 100)]}
        if d[:, 4].max() > 0:  # -DOSA_PART_CLOCKS build: phases of the (last) segment of the plain workgroups, shader cycles
            pl = ~straddle
            rec['cycles_prologue_median'] = int(np.median((d[:, 4] - d[:, 1])[pl]))
            if d[:, 12].max() > 0:
                seq = [1, 8, 9, 10, 11, 12, 4]
                rec['prologue_marks_W1_mv_tables_gather_W23_barrier'] = [int(np.median((d[:, b] - d[:, a])[pl])) for a, b in zip(seq[:-1], seq[1:])]
            rec['cycles_first_chunk_median'] = int(np.median((d[:, 5] - d[:, 4])[pl]))
            rec['cycles_other_chunks_median'] = int(np.median((d[:, 6] - d[:, 5])[pl]))
            rec['cycles_epilogue_median'] = int(np.median((d[:, 3] - d[:, 6])[pl]))
        out['rows'].append(rec)
        print(json.dumps(rec), flush=True)
        del up, ac
    if args.out:
        json.dump(out, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
