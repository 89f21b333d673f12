# round 3: hunt for the intermittent failure of the wide-split data-parallel test inside a full-suite run
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
: > gpurun_out/r3_flaky3.log
for i in 1 2 3 4 5 6; do
  timeout 400 python -m pytest tests/test_buffer_gpu.py tests/test_dp_gpu.py tests/test_host_env_gpu.py tests/test_mlp_gpu.py -q -m gpu -x --tb=short --show-capture=no --durations=5 2>&1 | tail -40 >> gpurun_out/r3_flaky3.log
  echo "=== loop $i done" >> gpurun_out/r3_flaky3.log
done
grep -n "passed\|failed\|Error\|assert" gpurun_out/r3_flaky3.log | head -60
