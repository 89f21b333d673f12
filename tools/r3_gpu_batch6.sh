set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_buffer_gpu.py -q -x -m gpu 2>&1 | tail -15 | tee gpurun_out/r3_b6_pytest.log
timeout 900 python tools/gae_bandwidth.py --only-gae 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_gae_bandwidth.log
