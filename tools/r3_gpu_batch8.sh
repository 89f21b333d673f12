# round 3: one-XCC placement A/B of the plain pass, rocprofv3 kernel stats and the PMC passes of the bench
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_mlp_gpu.py tests/test_config_shapes_gpu.py -x -q -m gpu -k "pass or ppolag_update or config" 2>&1 | tail -3
for i in 1 2 3; do
  for v in 0 1; do
    OSA_PASS_ONE_XCC=$v timeout 300 python bench.py --no-cpu-baseline --no-variant --steps 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one_xcc=$v', d['value'], d['roofline']['us_per_optimiser_step'])"
  done
done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r3_prof $R/gpurun_out/r3_pmc_*
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_prof -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r3_prof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/r3_pmc_$c -- python $R/bench.py --steps 1 --warmup 2 --update-iters 4 --no-cpu-baseline --no-variant > $R/gpurun_out/r3_pmc_$c.log 2>&1
  OSA_PASS_ONE_XCC=0 timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/r3_pmc_3xcc_$c -- python $R/bench.py --steps 1 --warmup 2 --update-iters 4 --no-cpu-baseline --no-variant > $R/gpurun_out/r3_pmc_3xcc_$c.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/r3_pmc_variant_$c -- python $R/bench.py --steps 1 --warmup 3 --batch-size 16384 --update-iters 8 --no-cpu-baseline --no-variant > $R/gpurun_out/r3_pmc_variant_$c.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/r3_pmc_sq -- python $R/bench.py --steps 1 --warmup 2 --update-iters 2 --no-cpu-baseline --no-variant > $R/gpurun_out/r3_pmc_sq.log 2>&1
ls $R/gpurun_out/r3_prof/*/* | head -5
