# round 4: balanced partial gradients of the large-batch step: parity, A/B timing, per-kernel stats, bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu -k "balanced or large_batch or multiblock or update_graph or full_size or ppolag_update" 2>&1 | tail -8
timeout 600 python -m pytest tests/test_dp_golden_gpu.py tests/test_rccl_gpu.py -x -q -m gpu -k "largebatch or large_batch" 2>&1 | tail -3
for v in 1 0; do
  echo "== OSA_LARGE_BATCH_BALANCED=$v"
  OSA_LARGE_BATCH_BALANCED=$v timeout 300 python tools/large_batch_step_timing.py --out gpurun_out/r4_large_batch_step_bal$v.json 2>&1 | grep -v amdgpu | tail -5
done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r4_prof_lb
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_prof_lb -- python $R/tools/large_batch_step_timing.py > /dev/null 2>&1
f=$(find $R/gpurun_out/r4_prof_lb -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-220
cd $R
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r4_b3_bench.json 2> gpurun_out/r4_b3_bench.err
python -c "
import json;d=json.load(open('gpurun_out/r4_b3_bench.json'));v=d['throughput_variant'];print(d['value'],d['ms_per_step'],v['value'],v['ms_per_step'],v['roofline']['us_per_optimiser_step'])"
