"""GPU tool: where the epoch of the large-batch variant (batch_size 16 384, update_iters 8; bench.py's
`throughput_variant`) goes.  Two modes:

    rocprofv3 --kernel-trace -d gpurun_out/r3_variant_trace -o run -- python tools/variant_timeline.py --run
    python tools/variant_timeline.py --analyse gpurun_out/r3_variant_trace/*_kernel_trace.csv [--out profiles/...json]

--run: 3 warm-up epochs (eager, capture, replay), a marker kernel, then 6 epochs, a marker kernel.
--analyse: kernels between the markers, per epoch: device-busy time per kernel name, idle time between kernels.
"""
import argparse
import csv
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EPOCHS = 6


def run():
    import types

    import torch

    import bench

    args = types.SimpleNamespace(envs=4096, steps_per_env=16, kl_early_stop=False, algo='PPOLag', hidden_sizes=[64, 64])
    with tempfile.TemporaryDirectory() as d:
        algo = bench.make_algo(args, 1, 16384, 8, 3 + EPOCHS + 1, d)
        sync = lambda: torch.cuda.synchronize()  # noqa: E731
        bench.run_epochs(algo, 3, sync)
        mark = torch.zeros(7, device='cuda:0', dtype=torch.float64)
        mark.erfinv_()  # marker: the only erfinv of the process
        sync()
        import time
        t0 = time.perf_counter()
        bench.run_epochs(algo, EPOCHS, sync)
        dt = time.perf_counter() - t0
        mark.erfinv_()
        sync()
        print(json.dumps({'ms_per_epoch_wall': dt / EPOCHS * 1e3, 'env_steps_per_s': 65536 * EPOCHS / dt}))


def short(name):
    for key in ('osa_', 'rocprim', 'at::native::'):
        i = name.find(key)
        if i >= 0:
            name = name[i:]
            break
    return name.split('(')[0].split('<')[0][:60]


def analyse(path, out):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if 'erfinv' in r[2].lower()]
    if len(marks) < 2:
        raise SystemExit(f'markers not found ({len(marks)})')
    seg = rows[marks[0] + 1: marks[-1]]
    t_begin, t_end = rows[marks[0]][1], rows[marks[-1]][0]
    busy, count = {}, {}
    idle_gaps = []
    gap_sites = {}
    last_end, last_name = t_begin, 'marker'
    for s, e, n in seg:
        k = short(n)
        busy[k] = busy.get(k, 0) + (e - s)
        count[k] = count.get(k, 0) + 1
        if s > last_end:
            idle_gaps.append(s - last_end)
            if s - last_end > 20000:
                site = f'{last_name} -> {k}'
                g = gap_sites.setdefault(site, [0, 0])
                g[0] += 1
                g[1] += s - last_end
        if e >= last_end:
            last_end, last_name = e, k
    total = t_end - t_begin
    busy_total = sum(busy.values())
    rep = {'epochs': EPOCHS, 'us_per_epoch_device_span': total / EPOCHS / 1e3,
           'us_per_epoch_kernels': busy_total / EPOCHS / 1e3,
           'us_per_epoch_idle': sum(idle_gaps) / EPOCHS / 1e3,
           'idle_gaps_per_epoch': len(idle_gaps) / EPOCHS,
           'idle_gaps_over_20us_per_epoch': sum(1 for g in idle_gaps if g > 20000) / EPOCHS,
           'us_in_gaps_over_20us_per_epoch': sum(g for g in idle_gaps if g > 20000) / EPOCHS / 1e3,
           'gaps_over_20us': {k: {'per_epoch': v[0] / EPOCHS, 'us_per_epoch': v[1] / EPOCHS / 1e3}
                              for k, v in sorted(gap_sites.items(), key=lambda kv: -kv[1][1])},
           'kernels': {k: {'launches_per_epoch': count[k] / EPOCHS, 'us_per_epoch': busy[k] / EPOCHS / 1e3,
                           'us_per_launch': busy[k] / count[k] / 1e3}
                       for k in sorted(busy, key=lambda k: -busy[k])}}
    print(json.dumps(rep, indent=1))
    if out:
        json.dump(rep, open(out, 'w'), indent=1)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--run', action='store_true')
    ap.add_argument('--analyse', default='')
    ap.add_argument('--out', default='')
    a = ap.parse_args()
    if a.run:
        run()
    else:
        analyse(a.analyse, a.out)
