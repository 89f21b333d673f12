#!/usr/bin/env python
"""TEST INFRASTRUCTURE (not product code).  Per-epoch logged statistics of the UNMODIFIED reference trained on
the learning-parity CMDP (SynthReach-v0, the configuration of tests/golden/learning_reach.json): every Misc/*,
Train/*, Loss/*, Metrics/*, Value/* column of the reference logger's progress.csv, per seed -> one JSON.  The
counterpart of tools/algo_epoch_stats.py (which dumps the same columns of the HIP path).

    OMP_NUM_THREADS=1 python oracle/ref_epoch_stats.py CPO 0 20 profiles/r2_cpo_epoch_stats_reference.json
"""
import csv
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402
import ref_harness  # noqa: E402

algo, lo, hi, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
ref_harness.import_reference()
import torch  # noqa: E402

torch.set_num_threads(1)
import omnisafe  # noqa: E402
from omnisafe.utils.config import get_default_kwargs_yaml  # noqa: E402

ref_harness.register_reach_env()
c = make_golden.LEARNING_CFG
res = {}
for seed in range(lo, hi):
    d = tempfile.mkdtemp()
    defaults = get_default_kwargs_yaml(algo, c['env_id'], 'on-policy').todict()
    cfg = make_golden.learning_custom_cfgs(algo, seed, 'cpu', d, defaults)
    cfg['train_cfgs']['torch_threads'] = 1
    cfg['logger_cfgs'].update({'use_wandb': False, 'use_tensorboard': False})
    omnisafe.Agent(algo, c['env_id'], custom_cfgs=cfg).learn()
    for root, _, files in os.walk(d):
        if 'progress.csv' in files:
            rows = list(csv.DictReader(open(os.path.join(root, 'progress.csv'))))
    res[str(seed)] = {k: [float(r[k]) for r in rows] for k in rows[0]
                      if k.split('/')[0] in ('Misc', 'Train', 'Loss', 'Metrics', 'Value')}
    json.dump(res, open(out, 'w'))
print('wrote', out)
