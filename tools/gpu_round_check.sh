set -x
cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
timeout 600 python bench.py > gpurun_out/bench_r1_e.json 2> gpurun_out/bench_r1_e.err; tail -c 2500 gpurun_out/bench_r1_e.json
OSA_DIST_BACKEND=gloo OSA_SINGLE_DEVICE_RANKS=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/bench_r1_2rank_1gpu.json 2> gpurun_out/bench_r1_2rank.err; tail -c 1500 gpurun_out/bench_r1_2rank_1gpu.json; tail -3 gpurun_out/bench_r1_2rank.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_r1e -- python /root/repo/bench.py --steps 2 --warmup 1 > /root/repo/gpurun_out/prof_r1e.log 2>&1
ls -R /root/repo/gpurun_out/prof_r1e | head
