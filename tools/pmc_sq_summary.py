#!/usr/bin/env python
"""SQ counter pass (rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY; one pass, no other trace domain)
-> profiles/r2_pmc_sq_mfma_busy.{md,json}: per kernel MFMA instructions, MFMA-busy cycles and the fraction of the
kernel's wave-time the matrix pipe was busy (SURVEY.md 8d "MFMA busy").

    python tools/pmc_sq_summary.py gpurun_out/pmc_sq [profiles/r2_pmc_sq_mfma_busy]

Units (MI355X_MICROARCH.md, PMC table): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles (32 per
v_mfma_f32_16x16x4_f32); SQ_WAVE_CYCLES / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_VALU count quad-cycles (x 4 = cycles)
summed over waves."""
import collections
import csv
import glob
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_sq'
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                        'profiles', 'r2_pmc_sq_mfma_busy')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
        if not name.startswith(('osa_', 'gm_')):
            continue
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
            dur[name].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
            agg[name]['_waves'].append(float(r['Grid_Size']) / 64.0)
res = {}
for k, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    wave_cycles = 4.0 * m.get('SQ_WAVE_CYCLES', 0.0)
    waves = m.get('_waves', 1.0)
    res[k] = {'launches': len(v['SQ_WAVE_CYCLES']), 'us_per_launch_under_pmc': round(sum(dur[k]) / len(dur[k]), 2),
              'waves_per_launch': waves, 'mfma_instructions': m.get('SQ_INSTS_MFMA', 0.0),
              'mfma_flops_mops_f32': m.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0.0) * 512,
              'valu_instructions': m.get('SQ_INSTS_VALU', 0.0), 'mfma_busy_cycles': m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0),
              'wave_cycles': wave_cycles,
              'mfma_busy_fraction_of_wave_time': round(m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / wave_cycles, 4) if wave_cycles else None,
              'issue_stall_fraction': round(4.0 * m.get('SQ_WAIT_INST_ANY', 0.0) / wave_cycles, 4) if wave_cycles else None}
json.dump(res, open(out + '.json', 'w'), indent=1)
with open(out + '.md', 'w') as f:
    f.write('# SQ counters per kernel (rocprofv3 --pmc, one pass; tools/pmc_sq_summary.py)\n\n'
            '`mfma busy` = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES): the share of the resident waves\' time '
            'during which their SIMD\'s matrix pipe executed an MFMA (each wave of these kernels has a SIMD to itself '
            'or shares it with at most one other).\n\n'
            '| kernel | launches | waves | MFMA instr / launch | VALU instr / launch | mfma busy | issue stall |\n|---|---|---|---|---|---|---|\n')
    for k, r in sorted(res.items(), key=lambda kv: -kv[1]['mfma_busy_cycles']):
        f.write(f"| `{k}` | {r['launches']} | {r['waves_per_launch']:.0f} | {r['mfma_instructions']:.0f} | "
                f"{r['valu_instructions']:.0f} | {r['mfma_busy_fraction_of_wave_time']} | {r['issue_stall_fraction']} |\n")
print(open(out + '.md').read())
