"""Turns the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) into
<out_base>.{json,_table.md} (third argument): HBM-side bytes per launch of every libomnisafe_amd kernel.

    python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE

Units / corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM + rocprofv3 PMC sections): the
counter values are KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane) reads, so
fetch bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is used as reported."""
import csv
import glob
import json
import os
import sys


def load(d, counter):
    out = {}
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] != counter:
                continue
            name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
            if not name.startswith(('osa_', 'gm_', 'gs_')):
                continue
            e = out.setdefault(name, [0, 0.0])
            e[0] += 1
            e[1] += float(r['Counter_Value'])
    return out


def main(fetch_dir, write_dir, out_base):
    f, w = load(fetch_dir, 'FETCH_SIZE'), load(write_dir, 'WRITE_SIZE')
    res = {}
    for k in f:
        n = f[k][0]
        fk = f[k][1] / n
        wk = (w.get(k, [1, 0.0])[1] / max(w.get(k, [1, 0.0])[0], 1))
        res[k] = {'launches': n, 'fetch_size_kib_raw': round(fk, 1), 'write_size_kib': round(wk, 1),
                  'traffic_bytes_per_launch': int((2 * fk + wk) * 1024)}
    # the digest of the kernel sources this was measured on: bench.py ignores the file once the kernels change
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from omnisafe_amd import build as _b

    out = dict(res, _abi_digest=_b.source_digest())
    json.dump(out, open(out_base + '.json', 'w'), indent=1)
    lines = ['| kernel | launches | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | traffic bytes / launch (2xFETCH + WRITE) |',
             '|---|---|---|---|---|']
    for k, v in res.items():
        lines.append(f"| `{k}` | {v['launches']} | {v['fetch_size_kib_raw']} | {v['write_size_kib']} | "
                     f"{v['traffic_bytes_per_launch'] / 1e6:.2f} MB |")
    open(out_base + '_table.md', 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else os.path.join(root, 'profiles', 'r2_pmc_traffic'))
