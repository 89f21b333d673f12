#!/bin/bash
# Copies what tools/gpu_round_check.sh left under gpurun_out/ into profiles/ (tracked): run HERE after the gpurun call.
#   tools/collect_round.sh r6
set -e
cd "$(dirname "$0")/.."
T=${1:-r6}; O=gpurun_out; P=profiles
for f in bench_final.json final_pytest.log rocprofv3_kernel_stats_bench.csv bench_CPO.json bench_TRPOLag.json \
         rocprofv3_kernel_stats_CPO.csv rocprofv3_kernel_stats_TRPOLag.csv bench_hidden1024.json general_mlp_timing.json \
         general_mlp_timing_tiled_B64.json rocprofv3_kernel_stats_general_1024_B64.csv \
         rocprofv3_kernel_stats_general_1024_B16384.csv gae_bandwidth.json gae_bandwidth.md large_batch_step.json \
         part_timeline.json dp_shapes_timing.json allreduce_step_world1.json rocprofv3_kernel_stats_allreduce_world1.csv \
         reference_full_epoch_config2.json pmc_traffic_general_1024_B64.json pmc_traffic_general_1024_B64_table.md \
         p2p_timing.json skinny_probe.txt fvp_phase_clocks.txt variant_timeline.json rccl_world1_timing.json; do
  [ -f $O/${T}_$f ] && cp $O/${T}_$f $P/${T}_$f || echo "missing ${T}_$f"
done
for t in bench variant CPO TRPOLag; do cp $O/${T}_pmc_traffic_$t.json $O/${T}_pmc_traffic_${t}_table.md $P/; done
for t in bench variant CPO TRPOLag general_1024; do cp $O/${T}_pmc_sq_mfma_busy_$t.json $O/${T}_pmc_sq_mfma_busy_$t.md $P/; done
cp $O/${T}_bench_2ranks_on_1gpu.json $P/${T}_bench_2ranks_on_1gpu_codepath_check.json
# (tools/baseline_configs.py writes fixed names)
[ -f $O/r4_baseline_configs.json ] && cp $O/r4_baseline_configs.json $P/${T}_baseline_configs.json && cp $O/r4_baseline_configs.md $P/${T}_baseline_configs.md
[ -f $O/${T}_algo_sweep.txt ] && cp $O/${T}_algo_sweep.txt $P/${T}_algo_sweep.md
python - <<PY
import json
from omnisafe_amd import build
d = build.source_digest()
for t in ('bench', 'variant', 'CPO', 'TRPOLag', 'general_1024_B64'):
    got = json.load(open('$P/${T}_pmc_traffic_%s.json' % t)).get('_abi_digest')
    print('traffic', t, 'digest', 'matches the working tree' if got == d else 'STALE (%s v %s)' % (got, d))
PY
