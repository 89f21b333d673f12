R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dp_golden_gpu.py tests/test_rccl_gpu.py -q -m gpu -s --tb=short 2>&1 | grep -v "amdgpu.ids" | tail -150 > gpurun_out/r4_b2_dp.log; tail -70 gpurun_out/r4_b2_dp.log
