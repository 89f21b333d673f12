# End-of-round measurement set of round 3 on one MI355X (run through gpurun): the GPU suite, the default bench, the
# 2-rank code-path check, the BASELINE configs on one GPU, GAE bandwidth, single-GPU pass timings.
# (rocprofv3 kernel stats + PMC passes of the same bench command: tools/r3_gpu_batch8.sh; DP shapes: tools/dp_shapes_timing.py)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu --tb=short --show-capture=no 2>&1 | tail -120 > gpurun_out/r3_final_pytest.log; tail -4 gpurun_out/r3_final_pytest.log
timeout 600 python bench.py > gpurun_out/r3_bench_final.json 2> gpurun_out/r3_bench_final.err; tail -c 800 gpurun_out/r3_bench_final.json
OSA_DIST_BACKEND=gloo OSA_SINGLE_DEVICE_RANKS=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 2 > gpurun_out/r3_bench_2ranks_on_1gpu.json 2> gpurun_out/r3_bench_2rank.err; tail -c 300 gpurun_out/r3_bench_2ranks_on_1gpu.json
timeout 900 python tools/baseline_configs.py --no-reference > gpurun_out/r3_baseline_configs.log 2>&1; tail -8 gpurun_out/r3_baseline_configs.log
timeout 900 python tools/gae_bandwidth.py --out gpurun_out/r3_gae_bandwidth > gpurun_out/r3_gae_bandwidth.log 2>&1; tail -2 gpurun_out/r3_gae_bandwidth.log
timeout 300 python tools/wide_pass_timing.py 65536 > gpurun_out/r3_wide_pass_timing.log 2>&1; tail -12 gpurun_out/r3_wide_pass_timing.log
timeout 300 python tools/chunked_pass_timing.py > gpurun_out/r3_chunked_pass_timing.log 2>&1; tail -10 gpurun_out/r3_chunked_pass_timing.log
