#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -rfE --timeout 900 > gpurun_out/r2_gpu_tests_7.log 2>&1
grep -E "passed|failed|error" gpurun_out/r2_gpu_tests_7.log | tail -3
grep -E "^(FAILED|ERROR)" gpurun_out/r2_gpu_tests_7.log | head -30
python tools/algo_epoch_stats.py CPO 20 gpurun_out/r2_epoch_stats_CPO.json > /dev/null 2>&1; ls -la gpurun_out/r2_epoch_stats_CPO.json
python tools/wide_pass_timing.py 65536 2>&1 | grep -v amdgpu
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/r2_bench_prof.json 2> $R/gpurun_out/r2_bench_prof.err
find $R/gpurun_out/prof_bench -name "*kernel_stats.csv" | head -2
cd $R
python bench.py > gpurun_out/r2_bench_7.json 2> gpurun_out/r2_bench_7.err; python -c "
import json; d=json.load(open('gpurun_out/r2_bench_7.json')); print(d['value'], d['ms_per_step'], d['roofline']['us_per_optimiser_step'], d['throughput_variant']['value'], d['throughput_variant']['ms_per_step'], d['cpu_baseline']['value'], d.get('cpu_baseline_port',{}).get('value'))"
