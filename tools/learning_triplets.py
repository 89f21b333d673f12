#!/usr/bin/env python
"""The north_star's literal learning criterion as a NUMBER: "episode return/cost within +-1 sigma of reference
over 3 seeds", evaluated over many disjoint seed triplets instead of one.

    python tools/learning_triplets.py train  [ALGO ...] [--seeds 60]      (GPU: trains, dumps the curves)
    python tools/learning_triplets.py report [--out profiles/r2_learning_triplets]   (CPU: tables)

For every algorithm: the tail metric (mean of the last 3 epochs) of EpRet and EpCost, the reference's mean and
sigma over ITS seeds (tests/golden/learning_reach.json: 20 seeds per algorithm, 80 for CPO), then

  * pass rate of our disjoint triplets (seeds 0-2, 3-5, ...): |mean of 3 - mu_ref| <= sigma_ref;
  * the same pass rate for triplets of the REFERENCE's own seeds against the reference's leave-triplet-out mean
    and sigma -- the rate an exact re-implementation with other random streams would achieve (a 3-seed mean of
    a quantity with per-seed sigma s scatters by s / sqrt(3), so ~92 % for a Gaussian; less for the
    heavy-tailed episode cost);
  * the large-sample difference of means in units of its standard error.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')]
GOLDEN = os.path.join(ROOT, 'tests', 'golden', 'learning_reach.json')
KEYS = ('EpRet', 'EpCost')


def tails(curves, key, k):
    return np.array([np.mean(c[key][-k:]) for _, c in sorted(curves.items(), key=lambda kv: int(kv[0]))])


def triplet_rate(x, mu, sigma):
    n = len(x) // 3
    m = x[:3 * n].reshape(n, 3).mean(1)
    return float((np.abs(m - mu) <= sigma).mean()), n


def self_rate(r):
    """Reference triplets against the rest of the reference (leave-triplet-out mean and sigma)."""
    n = len(r) // 3
    ok = 0
    for i in range(n):
        rest = np.concatenate([r[:3 * i], r[3 * i + 3:]])
        ok += abs(r[3 * i:3 * i + 3].mean() - rest.mean()) <= rest.std(ddof=1)
    return ok / max(n, 1), n


def train(algos, n_seeds, out_dir):
    from test_learning_gpu import train_reach

    cfg = json.load(open(GOLDEN))['config']
    os.makedirs(out_dir, exist_ok=True)
    for algo in algos:
        ours = {str(s): train_reach(algo, s, cfg, tempfile.mkdtemp()) for s in range(n_seeds)}
        json.dump({'config': cfg, 'curves': ours}, open(os.path.join(out_dir, f'r2_learning_curves_{algo}.json'), 'w'))
        print(algo, n_seeds, 'seeds trained', flush=True)


def report(src_dirs, out):
    g = json.load(open(GOLDEN))
    k = g['config']['tail_epochs']
    rows = []
    for algo, ref in sorted(g['curves'].items()):
        ours = None
        for d in src_dirs:
            for name in (f'r2_learning_curves_{algo}.json', f'r1_learning_curves_{algo}.json'):
                p = os.path.join(d, name)
                if ours is None and os.path.exists(p):
                    ours = json.load(open(p))['curves']
        if ours is None or len(ours) < 6:
            continue
        for key in KEYS:
            r, o = tails(ref, key, k), tails(ours, key, k)
            mu, sigma = r.mean(), r.std(ddof=1)
            rate, n_tri = triplet_rate(o, mu, sigma)
            srate, sn = self_rate(r)
            se = np.sqrt(o.var(ddof=1) / len(o) + r.var(ddof=1) / len(r))
            rows.append({'algo': algo, 'metric': key, 'ref_seeds': len(r), 'ref_mean': mu, 'ref_sigma': sigma,
                         'our_seeds': len(o), 'our_mean': o.mean(), 'diff_in_sigma': (o.mean() - mu) / sigma,
                         'diff_in_se': (o.mean() - mu) / se, 'triplets': n_tri, 'triplet_pass_rate': rate,
                         'ref_self_triplets': sn, 'ref_self_pass_rate': srate})
    os.makedirs(os.path.dirname(out) or '.', exist_ok=True)
    json.dump(rows, open(out + '.json', 'w'), indent=1)
    with open(out + '.md', 'w') as f:
        f.write('# "within +-1 sigma of the reference over 3 seeds" as a pass rate over disjoint seed triplets\n\n'
                'Tail metric = mean of the last 3 of 10 epochs on SynthReach-v0; sigma over the reference\'s seeds '
                '(tools/learning_triplets.py).  "ref vs ref" = the same test applied to the reference\'s own seed '
                'triplets against the rest of its seeds: what an exact re-implementation with different random '
                'streams scores.\n\n'
                '| algo | metric | reference mean +- sigma (seeds) | ours mean (seeds) | diff / sigma | diff / SE | '
                'our triplets within 1 sigma | ref vs ref triplets within 1 sigma |\n|' + '---|' * 8 + '\n')
        for r in rows:
            f.write(f"| {r['algo']} | {r['metric']} | {r['ref_mean']:.3f} +- {r['ref_sigma']:.3f} ({r['ref_seeds']}) | "
                    f"{r['our_mean']:.3f} ({r['our_seeds']}) | {r['diff_in_sigma']:+.2f} | {r['diff_in_se']:+.2f} | "
                    f"{r['triplet_pass_rate'] * 100:.0f} % of {r['triplets']} | "
                    f"{r['ref_self_pass_rate'] * 100:.0f} % of {r['ref_self_triplets']} |\n")
    print(open(out + '.md').read())


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('cmd', choices=['train', 'report'])
    ap.add_argument('algos', nargs='*', default=['PPOLag', 'TRPOLag', 'CPO'])
    ap.add_argument('--seeds', type=int, default=60)
    ap.add_argument('--dir', default=os.path.join(ROOT, 'gpurun_out'))
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r2_learning_triplets'))
    a = ap.parse_args()
    if a.cmd == 'train':
        train(a.algos, a.seeds, a.dir)
    else:
        report([a.dir, os.path.join(ROOT, 'profiles')], a.out)
