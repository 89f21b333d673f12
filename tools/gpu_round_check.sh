# End-of-round measurement set on one MI355X (run through gpurun): GPU tests, default bench, 2-rank code-path
# check, rocprofv3 kernel stats of the default bench, the PMC passes (separate runs, --kernel-trace only:
# FETCH_SIZE, WRITE_SIZE, one SQ pass), the same two traffic passes for the buffer kernels at M = 16 M.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 600 gpurun_out/bench_final.json
OSA_DIST_BACKEND=gloo OSA_SINGLE_DEVICE_RANKS=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 2 > gpurun_out/bench_2rank_1gpu.json 2> gpurun_out/bench_2rank.err; tail -c 400 gpurun_out/bench_2rank_1gpu.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_final $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE $R/gpurun_out/pmc_sq $R/gpurun_out/pmc_gae_FETCH_SIZE $R/gpurun_out/pmc_gae_WRITE_SIZE
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python $R/bench.py --steps 2 --warmup 2 > $R/gpurun_out/prof_final.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python $R/bench.py --steps 1 --warmup 2 --update-iters 4 --no-cpu-baseline --no-variant > $R/gpurun_out/pmc_$c.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_gae_$c -- python $R/tools/gae_bandwidth.py --pmc-run --shapes 16,1048576 4096,4096 > $R/gpurun_out/pmc_gae_$c.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_sq -- python $R/bench.py --steps 1 --warmup 2 --update-iters 2 --no-cpu-baseline --no-variant > $R/gpurun_out/pmc_sq.log 2>&1
ls $R/gpurun_out/prof_final/* $R/gpurun_out/pmc_FETCH_SIZE/* $R/gpurun_out/pmc_gae_FETCH_SIZE/* | head
# wide-observation passes (config 4) and the BASELINE configs end to end next to the unmodified reference
cd $R
timeout 300 python tools/wide_pass_timing.py 65536 > gpurun_out/wide_pass_timing.log 2>&1; tail -25 gpurun_out/wide_pass_timing.log
timeout 1500 python tools/baseline_configs.py > gpurun_out/baseline_configs.log 2>&1; tail -8 gpurun_out/baseline_configs.log
timeout 300 python tools/chunked_pass_timing.py > gpurun_out/chunked_pass_timing.log 2>&1; tail -16 gpurun_out/chunked_pass_timing.log
timeout 300 python tools/dp_timing.py 2>&1 | grep "^W=" > gpurun_out/dp_timing.log; cat gpurun_out/dp_timing.log
# per-kernel time of the BASELINE configs 3 and 4 (trust-region family with chunked critic passes; wide observations)
cd /tmp
for c in 3 4; do
  rm -rf $R/gpurun_out/prof_cfg$c
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg$c -- python $R/tools/config_epoch_profile.py $c 4 > $R/gpurun_out/prof_cfg$c.log 2>&1
  grep epoch $R/gpurun_out/prof_cfg$c.log | tail -2
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_sq_cfg4 -- python $R/tools/config_epoch_profile.py 4 2 > $R/gpurun_out/pmc_sq_cfg4.log 2>&1
cd $R
timeout 600 python tools/gae_bandwidth.py > gpurun_out/gae_bandwidth.log 2>&1; tail -2 gpurun_out/gae_bandwidth.log
