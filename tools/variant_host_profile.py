"""GPU tool: host-side profile (cProfile) of the large-batch variant's epoch loop: which Python functions keep the
device waiting between the rollout graph, the update launches and the logger."""
import cProfile
import os
import pstats
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

args = types.SimpleNamespace(envs=4096, steps_per_env=16, kl_early_stop=False, algo='PPOLag', hidden_sizes=[64, 64])
with tempfile.TemporaryDirectory() as d:
    algo = bench.make_algo(args, 1, 16384, 8, 64, d)
    sync = lambda: torch.cuda.synchronize()  # noqa: E731
    bench.run_epochs(algo, 3, sync)
    pr = cProfile.Profile()
    pr.enable()
    bench.run_epochs(algo, 30, sync)
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(45)
