R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu -k "balanced or large_batch or multiblock or update_graph" 2>&1 | tail -3
python tools/part_kernel_timeline.py 2>&1 | grep -v amdgpu | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print({k:d[k] for k in d if k in ('B','workgroups','tasks_per_workgroup','kernel_span_us','duration_us_plain_median','duration_us_straddling','cycles_median')})
"
OSA_LARGE_BATCH_BALANCED=1 timeout 300 python tools/large_batch_step_timing.py 2>&1 | grep -v amdgpu | tail -4
