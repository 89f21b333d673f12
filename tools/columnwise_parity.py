#!/usr/bin/env python
"""Column-by-column comparison of two per-epoch dumps of the logger's progress.csv (the unmodified reference:
oracle/ref_epoch_stats.py; the HIP path: tools/algo_epoch_stats.py) over many seeds: for every logged column
and epoch a Welch t test (means) and a Levene test (spreads) between the two populations of runs.

    python tools/columnwise_parity.py REF.json OURS.json [out.md]

With ~ 35 columns x 10 epochs x 2 tests, about 7 cells below p = 0.01 are expected by chance; a real difference
shows as a column that is significant over several consecutive epochs (the table lists every column with two or
more cells below 0.01, or any below 0.001).  /Min and /Max columns are skipped: the reference logs the MEAN in
both (distributed.py:388-391 reduces the vector element-wise, logger.py:366 averages it), omnisafe_amd logs the
extremes (profiles/HISTORY.md §8)."""
import json
import sys

import numpy as np
from scipy import stats


def load(path):
    d = json.load(open(path))
    return d.get('runs', d)


def main():
    ref, ours = load(sys.argv[1]), load(sys.argv[2])
    out = open(sys.argv[3], 'w') if len(sys.argv) > 3 else sys.stdout
    keys = sorted(k for k in next(iter(ref.values())) if k in next(iter(ours.values()))
                  and not k.endswith(('/Min', '/Max')))
    E = len(next(iter(ref.values()))[keys[0]])
    print(f'# Column-wise parity: {len(ref)} reference runs ({sys.argv[1]}) vs {len(ours)} HIP runs ({sys.argv[2]})\n',
          file=out)
    print('| column | epochs with Welch p < 0.01 | epochs with Levene p < 0.01 | reference mean (last epoch) | ours | '
          'reference sd | ours |\n|---|---|---|---|---|---|---|', file=out)
    cells = hits = 0
    flagged = []
    for k in keys:
        R = np.array([ref[s][k] for s in ref], dtype=float)
        O = np.array([ours[s][k] for s in ours], dtype=float)
        if R.shape[1] != O.shape[1] or np.allclose(R.std(0), 0) and np.allclose(O.std(0), 0):
            continue
        with np.errstate(all='ignore'):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                pt = np.nan_to_num([stats.ttest_ind(R[:, e], O[:, e], equal_var=False).pvalue for e in range(E)], nan=1.0)
                pl = np.nan_to_num([stats.levene(R[:, e], O[:, e]).pvalue for e in range(E)], nan=1.0)
        cells += 2 * E
        hits += int((pt < 0.01).sum() + (pl < 0.01).sum())
        if (pt < 0.01).sum() + (pl < 0.01).sum() >= 2 or (pt < 0.001).any() or (pl < 0.001).any():
            flagged.append(k)
        print(f'| {k} | {np.where(pt < 0.01)[0].tolist()} | {np.where(pl < 0.01)[0].tolist()} | {R[:, -1].mean():.4g} | '
              f'{O[:, -1].mean():.4g} | {R[:, -1].std(ddof=1):.3g} | {O[:, -1].std(ddof=1):.3g} |', file=out)
    print(f'\n{hits} of {cells} cells below p = 0.01 ({0.01 * cells:.1f} expected by chance); columns with >= 2 '
          f'cells below 0.01 or one below 0.001: {flagged or "none"}', file=out)


if __name__ == '__main__':
    main()
