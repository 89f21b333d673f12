# round 3, GPU call 2: the new data-parallel passes (split wide pass, chunked pass): equivalence tests, then timings
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_mlp_gpu.py -q -x -m gpu -k "data_parallel" 2>&1 | tail -40 | tee gpurun_out/r3_b2_pytest.log
timeout 900 python tools/dp_shapes_timing.py --out gpurun_out/r3_dp_shapes_timing.json 2>&1 | grep -v "^{" | tee gpurun_out/r3_dp_shapes_timing.log
