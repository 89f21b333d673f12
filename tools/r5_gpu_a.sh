cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_rccl_gpu.py tests/test_dp_golden_gpu.py tests/test_rollout_gpu.py -q -m gpu --tb=short --show-capture=no -x 2>&1 | tail -8 > $O/r5_a_pytest.log; tail -4 $O/r5_a_pytest.log
timeout 600 python bench.py > $O/r5_a_bench.json 2> $O/r5_a_bench.err; tail -c 1500 $O/r5_a_bench.json
OSA_SINGLE_DEVICE_RANKS=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 2 --no-cpu-baseline > $O/r5_a_bench2.json 2> $O/r5_a_bench2.err; tail -c 800 $O/r5_a_bench2.json; tail -3 $O/r5_a_bench2.err
