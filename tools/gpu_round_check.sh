# End-of-round measurement set on one MI355X (run through gpurun): GPU tests, default bench, 2-rank code-path
# check, rocprofv3 kernel stats of the default bench, and the two PMC passes (separate runs, kernel-trace only).
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 600 gpurun_out/bench_final.json
OSA_DIST_BACKEND=gloo OSA_SINGLE_DEVICE_RANKS=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/bench_2rank_1gpu.json 2> gpurun_out/bench_2rank.err; tail -c 400 gpurun_out/bench_2rank_1gpu.json
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_final /root/repo/gpurun_out/pmc_FETCH_SIZE /root/repo/gpurun_out/pmc_WRITE_SIZE
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_final -- python /root/repo/bench.py --steps 2 --warmup 1 > /root/repo/gpurun_out/prof_final.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /root/repo/gpurun_out/pmc_$c -- python /root/repo/bench.py --steps 1 --warmup 1 --update-iters 4 --no-cpu-baseline --no-variant > /root/repo/gpurun_out/pmc_$c.log 2>&1
done
ls /root/repo/gpurun_out/prof_final/* /root/repo/gpurun_out/pmc_FETCH_SIZE/* | head
