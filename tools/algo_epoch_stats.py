#!/usr/bin/env python
"""Per-epoch logged statistics (every Misc/*, Train/*, Loss/*, Metrics/* column of progress.csv) of one
algorithm on SynthReach-v0 for seeds 0..n-1, trained with the learning-parity configuration -> one JSON.

    python tools/algo_epoch_stats.py CPO 20 gpurun_out/r2_epoch_stats_CPO.json

Used to compare the DISTRIBUTION of discrete decisions (CPO optimisation case, accepted line-search step) and
of step sizes with the same dump of the unmodified reference, beyond the return / cost curves."""
import csv
import glob
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')]

from test_learning_gpu import GOLDEN, reach_custom_cfgs  # noqa: E402

algo = sys.argv[1] if len(sys.argv) > 1 else 'CPO'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, 'gpurun_out', f'r2_epoch_stats_{algo}.json')
import omnisafe_amd  # noqa: E402

cfg = json.load(open(GOLDEN))['config']
res = {}
for seed in range(n):
    d = tempfile.mkdtemp()
    omnisafe_amd.Agent(algo, cfg['env_id'], custom_cfgs=reach_custom_cfgs(algo, seed, cfg, d)).learn()
    rows = list(csv.DictReader(open(glob.glob(os.path.join(d, '*', '*', 'progress.csv'))[0])))
    res[str(seed)] = {k: [float(r[k]) for r in rows] for k in rows[0]
                      if k.split('/')[0] in ('Misc', 'Train', 'Loss', 'Metrics', 'Value')}
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, 'w'))
print('wrote', out)
