"""GPU tool: per-epoch EpRet of one sibling algorithm on SynthReach-v0 with many seeds, next to the reference's 20
golden curves (tests/golden/learning_reach.json): is a per-epoch deviation of the 8-seed test sampling noise?

    python tools/sibling_epoch_check.py PPOSaute 32
"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')):  # (tool of the test infrastructure)
    sys.path.insert(0, p)
from test_learning_gpu import GOLDEN, train_reach  # noqa: E402

algo, n = sys.argv[1], int(sys.argv[2])
g = json.load(open(GOLDEN))
cfg, ref = g['config'], g['curves'][algo]
with tempfile.TemporaryDirectory() as d:
    ours = [train_reach(algo, s, cfg, d) for s in range(n)]
out = {'algo': algo, 'seeds': n, 'epochs': []}
for key in ('EpRet', 'EpCost'):
    for e in range(cfg['epochs']):
        o = np.array([c[key][e] for c in ours])
        r = np.array([c[key][e] for c in ref.values()])
        se = float(np.sqrt(o.var(ddof=1) / len(o) + r.var(ddof=1) / len(r)))
        rec = {'key': key, 'epoch': e, 'ours': round(float(o.mean()), 4), 'ours_first8': round(float(o[:8].mean()), 4),
               'ref': round(float(r.mean()), 4), 'sigma_ref': round(float(r.std(ddof=1)), 4), 'se': round(se, 4),
               'diff_in_sigma': round(float((o.mean() - r.mean()) / r.std(ddof=1)), 2),
               'diff_in_se': round(float((o.mean() - r.mean()) / se), 2)}
        out['epochs'].append(rec)
        print(rec, flush=True)
json.dump(out, open(os.path.join(os.environ.get('OSA_OUT', 'gpurun_out'), f'r3_sibling_epochs_{algo}.json'), 'w'), indent=1)
