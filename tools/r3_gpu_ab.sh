# round 3: same-box A/B of pass-kernel variants: correctness of the working-tree kernel, then alternating timings
# usage: bash tools/r3_gpu_ab.sh <variant suffix> [...]   ("new" = the default build)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_mlp_gpu.py tests/test_config_shapes_gpu.py -x -q -m gpu -k "pass or ppolag_update or replicated or config" 2>&1 | tail -3
bash tools/ab_run.sh "$@"
