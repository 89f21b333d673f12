set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r3_b5_pytest.log
timeout 600 python bench.py --ref-sample-iters 2 > gpurun_out/r3_bench_mid.json 2> gpurun_out/r3_bench_mid.err; tail -c 1500 gpurun_out/r3_bench_mid.json
OSA_DIST_BACKEND=gloo OSA_SINGLE_DEVICE_RANKS=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 2 > gpurun_out/r3_bench_2rank_1gpu.json 2> gpurun_out/r3_bench_2rank.err; tail -c 600 gpurun_out/r3_bench_2rank_1gpu.json
