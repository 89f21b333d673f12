"""Train PPOLag (or argv[1]) on SynthReach-v0 with seeds 0..n-1 on the GPU and dump the per-epoch curves
next to the reference's (tests/golden/learning_reach.json) -> gpurun_out/learning_<algo>.json."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')]
import numpy as np  # noqa: E402

from test_learning_gpu import GOLDEN, train_reach  # noqa: E402

algo = sys.argv[1] if len(sys.argv) > 1 else 'PPOLag'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # first seed (other seeds = other env layouts)
g = json.load(open(GOLDEN))
cfg = g['config']
ours = {}
for seed in range(first, first + n):
    ours[str(seed)] = train_reach(algo, seed, cfg, tempfile.mkdtemp())
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump({'config': cfg, 'curves': ours}, open(os.path.join(ROOT, 'gpurun_out', f'learning_{algo}.json' if first == 0 else f'learning_{algo}_from{first}.json'), 'w'),
          indent=1)
ref = g['curves'].get(algo)
for key in ('EpRet', 'EpCost', 'LagrangeMultiplier'):
    if key not in next(iter(ours.values())):  # CPO has no multiplier
        continue
    o = np.array([c[key] for c in ours.values()])
    print(key, 'ours mean/std per epoch:', np.round(o.mean(0), 3).tolist(), np.round(o.std(0, ddof=1), 3).tolist())
    if ref:
        r = np.array([c[key] for c in ref.values()])
        print(key, 'ref  mean/std per epoch:', np.round(r.mean(0), 3).tolist(), np.round(r.std(0, ddof=1), 3).tolist())
