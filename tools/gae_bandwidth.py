#!/usr/bin/env python
"""HBM roofline of the buffer kernels ("report GB/s at M in {65 536, 1 M, 16 M} to show the
curve"): times osa_gae_scan (lane-per-env, bit-exact), osa_gae_scan_tiled (time-parallel wavefront scan with
LDS staging) and the get() chain (osa_adv_stats_phase1/2 + osa_buffer_get) with HIP events on torch's stream.

    python tools/gae_bandwidth.py [--out gpurun_out/r2_gae_bandwidth] [--shapes T,N T,N ...] [--only-gae]

Algorithmic bytes per transition: GAE scan 36 B (reads r, c, v_r, v_c; writes adv_r, adv_c,
tgt_r, tgt_c, disc_ret; the 1-byte path-end flag and the bootstraps at path ends add 1-2 B and are not
counted), get() = 12 B statistics (adv_r twice, adv_c once) + 48 B for the six transposed scalar arrays +
8 (D_o + D_a) B for the transposed rows.  Denominators: 8 TB/s (MI355X_MICROARCH.md HBM peak) and the copy
bandwidth measured here with a 1 GiB device-to-device copy.  `--pmc-run` executes only a few repetitions of
the largest shape so that `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE` (separate passes) attributes traffic.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DEFAULT_SHAPES = [(16, 4096), (16, 65536), (16, 1 << 20),        # rollout-shaped: T = 16 (BASELINE config 2)
                  (256, 256), (256, 4096), (256, 65536),         # M = 65 536 / 1 M / 16 M at T = 256
                  (4096, 16), (4096, 256), (4096, 4096),         # long horizon
                  (5000, 4)]                                     # BASELINE config 1 (4 envs, T = 5000)
D_O, D_A = 60, 2


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'r3_gae_bandwidth'))
    ap.add_argument('--shapes', nargs='*')
    ap.add_argument('--only-gae', action='store_true')
    ap.add_argument('--pmc-run', action='store_true')
    args = ap.parse_args()
    from omnisafe_amd.buffer import VectorOnPolicyBuffer
    from omnisafe_amd.spaces import Box

    shapes = [tuple(int(x) for x in s.split(',')) for s in args.shapes] if args.shapes else DEFAULT_SHAPES
    if args.pmc_run and not args.shapes:
        shapes = [s for s in shapes if s[0] * s[1] >= (1 << 24)] or shapes[-1:]
    dev = torch.device('cuda:0')
    # copy bandwidth of this box (read + write bytes / time)
    src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    copy_us = timed(lambda: dst.copy_(src), 10)
    copy_gbps = 2 * src.numel() * 4 / copy_us / 1e3
    del src, dst
    torch.cuda.empty_cache()
    rows = []
    for T, N in shapes:
        M = T * N
        rng = np.random.default_rng(T + N)
        bufs = {}
        for variant in ('sequential', 'tiled', 'chained'):
            buf = VectorOnPolicyBuffer(Box(-np.inf, np.inf, (D_O,)), Box(-1, 1, (D_A,)), size=T, gamma=0.99,
                                       lam=0.95, lam_c=0.95, advantage_estimator='gae', penalty_coefficient=0.0,
                                       standardized_adv_r=True, standardized_adv_c=True, num_envs=N, device=dev,
                                       gae_variant=variant)
            if not bufs:
                for k in ('reward', 'value_r', 'value_c'):
                    buf.data[k].normal_()
                buf.data['cost'].copy_((torch.rand(T, N, device=dev) < 0.05).float())
                pe = torch.rand(T, N, device=dev) < (0.02 if T >= 64 else 0.0)
                pe[-1] = True
                buf.data['path_end'].copy_(pe.to(torch.uint8))
                buf.data['boot_r'].copy_(torch.where(pe, torch.randn(T, N, device=dev), 0.0))
                buf.data['boot_c'].copy_(torch.where(pe, torch.randn(T, N, device=dev), 0.0))
                buf.data['obs'].normal_()
                buf.data['act'].normal_()
            else:
                first = bufs['sequential']
                buf.data = first.data  # same inputs, same output arrays (results overwritten in turn)
            bufs[variant] = buf
        reps = 3 if args.pmc_run else max(5, min(200, int(2e8 // max(M, 1))))
        row = {'T': T, 'N': N, 'M': M}
        for variant, buf in bufs.items():
            buf.ptr = T
            us = timed(buf.compute_advantages, reps)
            row[f'gae_{variant}_us'] = round(us, 2)
            row[f'gae_{variant}_GBps'] = round(36.0 * M / us / 1e3, 1)
        # agreement of the two kernels on this shape (float32 outputs)
        bufs['sequential'].compute_advantages()
        a = bufs['sequential'].data['adv_r'].clone()
        bufs['tiled'].compute_advantages()
        b = bufs['tiled'].data['adv_r']
        row['tiled_vs_sequential_max_rel'] = float(((a - b).abs() / a.abs().clamp_min(1e-6)).max())
        row['tiled_bit_identical_frac'] = float((a == b).float().mean())
        bufs['chained'].compute_advantages()
        b = bufs['chained'].data['adv_r']
        row['chained_vs_sequential_max_rel'] = float(((a - b).abs() / a.abs().clamp_min(1e-6)).max())
        row['chained_bit_identical_frac'] = float((a == b).float().mean())
        row['auto_picks'] = VectorOnPolicyBuffer.gae_variant_for(T, N, 0)
        if not args.only_gae:
            buf = bufs['sequential']
            pe_keep = buf.data['path_end'].clone()

            def get():
                buf.ptr = T
                buf.data['path_end'].copy_(pe_keep)
                buf.get()

            def restore_only():
                buf.ptr = T
                buf.data['path_end'].copy_(pe_keep)
                buf.data['path_end'].zero_()

            us_all = timed(get, reps)
            us_gae = row['gae_sequential_us']
            us_misc = timed(restore_only, reps)
            us_get = max(us_all - us_gae - us_misc, 1e-3)  # statistics + transposition (6 launches)
            bytes_get = 12 + 48 + 8 * (D_O + D_A)
            row['get_chain_us'] = round(us_get, 2)
            row['get_chain_GBps'] = round(bytes_get * M / us_get / 1e3, 1)
        rows.append(row)
        print(json.dumps(row), flush=True)
        del bufs
        torch.cuda.empty_cache()
    res = {'device': torch.cuda.get_device_name(0), 'copy_bandwidth_GBps_1GiB_d2d': round(copy_gbps, 1),
           'hbm_peak_GBps': 8000.0, 'bytes_per_transition': {'gae_scan': 36, 'get_chain': 12 + 48 + 8 * (D_O + D_A)},
           'rows': rows}
    if args.pmc_run:
        return
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out + '.json', 'w'), indent=1)
    with open(args.out + '.md', 'w') as f:
        f.write(f'# GAE scan / get() bandwidth on {res["device"]} (tools/gae_bandwidth.py)\n\n')
        f.write(f'Measured device-to-device copy bandwidth (1 GiB, read + write): {copy_gbps:.0f} GB/s; HBM peak '
                '8000 GB/s.  GAE: 36 B / transition (SURVEY.md 8d); get() chain (2 statistics phases + fused '
                f'standardise/transpose of 6 scalar arrays + obs/act rows, D_o = {D_O}, D_a = {D_A}): '
                f'{12 + 48 + 8 * (D_O + D_A)} B / transition.\n\n')
        f.write('| T | N | M | lane-per-env us | GB/s | % of 8 TB/s | tiled us | GB/s | % of 8 TB/s | chained us | GB/s | '
                '% of 8 TB/s | auto picks | tiled == sequential (frac bit-identical, max rel) | chained == sequential | '
                'get() us | GB/s |\n|' + '---|' * 17 + '\n')
        for r in rows:
            f.write(f"| {r['T']} | {r['N']} | {r['M']} | {r['gae_sequential_us']} | {r['gae_sequential_GBps']} | "
                    f"{r['gae_sequential_GBps'] / 80:.1f} | {r['gae_tiled_us']} | {r['gae_tiled_GBps']} | "
                    f"{r['gae_tiled_GBps'] / 80:.1f} | {r['gae_chained_us']} | {r['gae_chained_GBps']} | "
                    f"{r['gae_chained_GBps'] / 80:.1f} | {r['auto_picks']} | {r['tiled_bit_identical_frac']:.6f}, "
                    f"{r['tiled_vs_sequential_max_rel']:.1e} | {r['chained_bit_identical_frac']:.6f}, "
                    f"{r['chained_vs_sequential_max_rel']:.1e} | {r.get('get_chain_us', '')} | "
                    f"{r.get('get_chain_GBps', '')} |\n")
    print('wrote', args.out + '.{json,md}')


if __name__ == '__main__':
    main()
