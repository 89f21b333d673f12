set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_mlp_gpu.py -q -x -m gpu -k "data_parallel or pass" 2>&1 | tail -5
timeout 900 python tools/dp_shapes_timing.py --worlds 1 2 4 8 --out gpurun_out/r3_dp_shapes_timing.json 2>&1 | grep -v "^{" | grep -v amdgpu
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-variant --steps 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['roofline']['us_per_optimiser_step'])"; done
