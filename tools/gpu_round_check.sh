# End-of-round measurement set on one MI355X (run through gpurun; ROUND tag as first argument, default r6).
#   stage 1  GPU suite, default bench, 2-rank code-path check (both ranks on the one GPU over gloo; incl. the
#            large-batch variant under data parallelism)
#   stage 2  rocprofv3 kernel stats of the bench command; FETCH_SIZE / WRITE_SIZE passes (separate runs, --kernel-trace
#            only) of the headline workload and of the large-batch variant -> profiles-ready JSON incl. the source
#            digest (tools/pmc_summary.py); one SQ pass each
#   stage 3  trust-region family: bench lines of CPO and TRPOLag (roofline_fvp), kernel stats, SQ pass
#   stage 4  general networks: bench line at hidden 1024 x 1024, kernel stats, SQ pass, timing table
#   stage 5  buffer kernels (GAE bandwidth), BASELINE configs on one GPU, pass timings
#   stage 6  per-step all-reduce mode over RCCL at world 1 (eager v graph, kernel split), every algorithm's epoch time
#   stage 7  one-shot peer exchange at 1 / 2 / 4 / 8 ranks on the one device, skinny-kernel probe, FVP phase clocks
# Everything lands under gpurun_out/<ROUND>_*; copy what is to be judged into profiles/.
set -x
T=${1:-r6}
STAGES=${2:-1234567}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
mkdir -p $O
SQ="SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
if [[ $STAGES == *1* ]]; then
timeout 1500 python -m pytest tests -q -m gpu --tb=short --show-capture=no 2>&1 | tail -60 > $O/${T}_final_pytest.log; tail -4 $O/${T}_final_pytest.log
timeout 600 python bench.py > $O/${T}_bench_final.json 2> $O/${T}_bench_final.err; tail -c 600 $O/${T}_bench_final.json
# (the driver's own command line: bench.py launches its two ranks itself; one device -> gloo, a code-path check)
OSA_SINGLE_DEVICE_RANKS=1 timeout 900 python bench.py --gpus 2 --steps 2 --warmup 2 --update-iters 8 --allreduce-update-iters 1 > $O/${T}_bench_2ranks_on_1gpu.json 2> $O/${T}_bench_2rank.err; tail -c 400 $O/${T}_bench_2ranks_on_1gpu.json
fi
cd /tmp && export TMPDIR=/tmp
if [[ $STAGES == *2* ]]; then
rm -rf $O/${T}_prof $O/${T}_pmc_*
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $O/${T}_prof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${T}_pmc_bench_$c -- python $R/bench.py --steps 1 --warmup 2 --update-iters 4 --no-cpu-baseline --no-variant > $O/${T}_pmc_bench_$c.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${T}_pmc_variant_$c -- python $R/bench.py --steps 1 --warmup 3 --batch-size 16384 --update-iters 8 --no-cpu-baseline --no-variant > $O/${T}_pmc_variant_$c.log 2>&1
done
python $R/tools/pmc_summary.py $O/${T}_pmc_bench_FETCH_SIZE $O/${T}_pmc_bench_WRITE_SIZE $O/${T}_pmc_traffic_bench | head -12
python $R/tools/pmc_summary.py $O/${T}_pmc_variant_FETCH_SIZE $O/${T}_pmc_variant_WRITE_SIZE $O/${T}_pmc_traffic_variant | head -12
timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/${T}_pmc_sq_bench -- python $R/bench.py --steps 1 --warmup 2 --update-iters 2 --no-cpu-baseline --no-variant > $O/${T}_pmc_sq_bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/${T}_pmc_sq_variant -- python $R/bench.py --steps 1 --warmup 3 --batch-size 16384 --update-iters 8 --no-cpu-baseline --no-variant > $O/${T}_pmc_sq_variant.log 2>&1
python $R/tools/pmc_sq_summary.py $O/${T}_pmc_sq_bench $O/${T}_pmc_sq_mfma_busy_bench | head -20
python $R/tools/pmc_sq_summary.py $O/${T}_pmc_sq_variant $O/${T}_pmc_sq_mfma_busy_variant | head -20
rm -rf $O/${T}_pmc_bench_* $O/${T}_pmc_variant_* $O/${T}_pmc_sq_bench $O/${T}_pmc_sq_variant
f=$(find $O/${T}_prof -name "*kernel_stats.csv" | head -1); cp $f $O/${T}_rocprofv3_kernel_stats_bench.csv; rm -rf $O/${T}_prof
fi
if [[ $STAGES == *3* ]]; then
for A in CPO TRPOLag; do
  cd $R; timeout 600 python bench.py --algo $A --batch-size 128 --update-iters 10 --no-cpu-baseline > $O/${T}_bench_$A.json 2> $O/${T}_bench_$A.err; tail -c 700 $O/${T}_bench_$A.json
  cd /tmp; rm -rf $O/${T}_prof_$A $O/${T}_pmc_sq_$A
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_$A -- python $R/bench.py --algo $A --batch-size 128 --update-iters 10 --steps 2 --warmup 2 --no-cpu-baseline > $O/${T}_prof_$A.log 2>&1
  f=$(find $O/${T}_prof_$A -name "*kernel_stats.csv" | head -1); cp $f $O/${T}_rocprofv3_kernel_stats_$A.csv; rm -rf $O/${T}_prof_$A
  timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/${T}_pmc_sq_$A -- python $R/bench.py --algo $A --batch-size 128 --update-iters 10 --steps 1 --warmup 2 --no-cpu-baseline > $O/${T}_pmc_sq_$A.log 2>&1
  python $R/tools/pmc_sq_summary.py $O/${T}_pmc_sq_$A $O/${T}_pmc_sq_mfma_busy_$A | head -16; rm -rf $O/${T}_pmc_sq_$A
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${T}_pmc_${A}_$c -- python $R/bench.py --algo $A --batch-size 128 --update-iters 2 --steps 1 --warmup 2 --no-cpu-baseline > $O/${T}_pmc_${A}_$c.log 2>&1
  done
  python $R/tools/pmc_summary.py $O/${T}_pmc_${A}_FETCH_SIZE $O/${T}_pmc_${A}_WRITE_SIZE $O/${T}_pmc_traffic_$A | grep -E "fvp|kernel \|" | head -6
  rm -rf $O/${T}_pmc_${A}_FETCH_SIZE $O/${T}_pmc_${A}_WRITE_SIZE
done
fi
if [[ $STAGES == *4* ]]; then
cd $R; timeout 900 python bench.py --hidden-sizes 1024 1024 --update-iters 1 --steps 2 --no-cpu-baseline > $O/${T}_bench_hidden1024.json 2> $O/${T}_bench_hidden1024.err; tail -c 900 $O/${T}_bench_hidden1024.json
timeout 600 python tools/general_mlp_timing.py --out $O/${T}_general_mlp_timing.json 2>&1 | grep -v amdgpu
OSA_GMLP_SKINNY=0 timeout 300 python tools/general_mlp_timing.py --shapes 1024x1024:64 512x512x512:64 256x128:64 --out $O/${T}_general_mlp_timing_tiled_B64.json 2>&1 | grep -v amdgpu
cd /tmp; rm -rf $O/${T}_prof_gm $O/${T}_pmc_sq_gm
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_gm -- python $R/tools/general_mlp_timing.py --shapes 1024x1024:64 --reps 10 > /dev/null 2>&1
f=$(find $O/${T}_prof_gm -name "*kernel_stats.csv" | head -1); cp $f $O/${T}_rocprofv3_kernel_stats_general_1024_B64.csv; rm -rf $O/${T}_prof_gm
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${T}_pmc_gm_$c -- python $R/tools/general_mlp_timing.py --shapes 1024x1024:64 --reps 3 > /dev/null 2>&1
done
python $R/tools/pmc_summary.py $O/${T}_pmc_gm_FETCH_SIZE $O/${T}_pmc_gm_WRITE_SIZE $O/${T}_pmc_traffic_general_1024_B64 | head -12
rm -rf $O/${T}_pmc_gm_FETCH_SIZE $O/${T}_pmc_gm_WRITE_SIZE
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_gm -- python $R/tools/general_mlp_timing.py --shapes 1024x1024:16384 --reps 5 > /dev/null 2>&1
f=$(find $O/${T}_prof_gm -name "*kernel_stats.csv" | head -1); cp $f $O/${T}_rocprofv3_kernel_stats_general_1024_B16384.csv; rm -rf $O/${T}_prof_gm
timeout 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/${T}_pmc_sq_gm -- python $R/tools/general_mlp_timing.py --shapes 1024x1024:16384 --reps 2 > /dev/null 2>&1
python $R/tools/pmc_sq_summary.py $O/${T}_pmc_sq_gm $O/${T}_pmc_sq_mfma_busy_general_1024 | head -16; rm -rf $O/${T}_pmc_sq_gm
fi
if [[ $STAGES == *5* ]]; then
cd $R
timeout 900 python tools/baseline_configs.py --no-reference > $O/${T}_baseline_configs.log 2>&1; tail -8 $O/${T}_baseline_configs.log
timeout 900 python tools/gae_bandwidth.py --out $O/${T}_gae_bandwidth > $O/${T}_gae_bandwidth.log 2>&1; tail -2 $O/${T}_gae_bandwidth.log
timeout 300 python tools/large_batch_step_timing.py --out $O/${T}_large_batch_step.json 2>&1 | grep -v amdgpu | tail -5
timeout 300 python tools/part_kernel_timeline.py --out $O/${T}_part_timeline.json 2>&1 | grep -v amdgpu | tail -6 | cut -c1-300
timeout 600 python tools/dp_shapes_timing.py --out $O/${T}_dp_shapes_timing.json 2>&1 | grep -v "^{" | grep -v amdgpu | tail -12
fi
if [[ $STAGES == *6* ]]; then
# per-step all-reduce mode at world 1 over RCCL: eager v captured graph, and the per-kernel split of a step
cd $R; timeout 300 python tools/allreduce_step_trace.py --out $O/${T}_allreduce_step_world1.json 2>&1 | tail -1
cd /tmp; rm -rf $O/${T}_prof_ar
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_ar -- python $R/tools/allreduce_step_trace.py --rows 16384 > /dev/null 2>&1
f=$(find $O/${T}_prof_ar -name "*kernel_stats.csv" | head -1); cp $f $O/${T}_rocprofv3_kernel_stats_allreduce_world1.csv; rm -rf $O/${T}_prof_ar
cp $O/rccl_world1_timing.json $O/${T}_rccl_world1_timing.json 2>/dev/null
cd $R; timeout 900 python tools/algo_sweep.py > $O/${T}_algo_sweep.txt 2>&1; tail -25 $O/${T}_algo_sweep.txt
# the unmodified reference with ALL 40 passes on this box's host (the bench line's cpu_baseline extrapolates from 4)
timeout 400 python oracle/ref_cpu_baseline.py --envs 4096 --steps-per-env 16 --batch-size 64 --update-iters 40 --sample-iters 40 --threads 16 2>/dev/null | tail -1 > $O/${T}_reference_full_epoch_config2.json; cat $O/${T}_reference_full_epoch_config2.json | cut -c1-300
fi
if [[ $STAGES == *7* ]]; then
# round 6: one-shot peer exchange (ranks on the one device), probes of the skinny kernels, phase clocks of the FVP chunk
# (build first, here on the CPU:  hipcc ... tools/skinny_probe.hip -DGS_CLOCKS -o tools/_probe/skinny_probe ;
#  tools/build_variant_lib.sh fvpclocks fvp_kernel.hip -DOFV_CLOCKS)
cd $R
timeout 600 python tools/p2p_timing.py --out $O/${T}_p2p_timing.json 2>&1 | grep '^{"shape' | cut -c1-240
[ -x tools/_probe/skinny_probe ] && tools/_probe/skinny_probe > $O/${T}_skinny_probe.txt 2>&1; tail -12 $O/${T}_skinny_probe.txt
[ -f omnisafe_amd/lib/libomnisafe_amd_fvpclocks.so ] && OSA_LIB_PATH=$R/omnisafe_amd/lib/libomnisafe_amd_fvpclocks.so timeout 300 python tools/fvp_phase_clocks.py --out $O/${T}_fvp_phase_clocks.txt 2>&1 | grep -v amdgpu | tail -15
fi
