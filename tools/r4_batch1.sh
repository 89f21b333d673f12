# round 4, first GPU call: the new data-parallel goldens, RCCL graph path, changed learning gate; then the whole suite + bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dp_golden_gpu.py tests/test_rccl_gpu.py -x -q -m gpu -s 2>&1 | grep -v "amdgpu.ids" | tail -40 > gpurun_out/r4_b1_dp.log; tail -30 gpurun_out/r4_b1_dp.log
timeout 1500 python -m pytest tests -q -m gpu --tb=short --show-capture=no --deselect tests/test_dp_golden_gpu.py --deselect tests/test_rccl_gpu.py 2>&1 | tail -40 > gpurun_out/r4_b1_pytest.log; tail -6 gpurun_out/r4_b1_pytest.log
timeout 600 python bench.py > gpurun_out/r4_b1_bench.json 2> gpurun_out/r4_b1_bench.err; tail -c 1500 gpurun_out/r4_b1_bench.json
