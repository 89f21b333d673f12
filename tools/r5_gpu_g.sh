cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_general_mlp_gpu.py -q -m gpu --tb=short --show-capture=no -k "skinny or oracle or general_widths" 2>&1 | tail -5 > $O/r5_g_pytest.log; tail -5 $O/r5_g_pytest.log | cut -c1-250
timeout 1200 python -m pytest tests/test_rccl_gpu.py tests/test_dp_golden_gpu.py tests/test_dp_gpu.py -q -m gpu --tb=short --show-capture=no 2>&1 | tail -25 > $O/r5_g_pytest_dp.log; tail -25 $O/r5_g_pytest_dp.log | cut -c1-250
timeout 300 python tools/general_mlp_timing.py --shapes 1024x1024:64 512x512x512:64 256x128:64 --reps 20 --out $O/r5_g_gm_timing.json 2>&1 | grep -v amdgpu | tail -3
timeout 300 python tools/allreduce_step_trace.py --out $O/r5_allreduce_step.json 2>&1 | tail -1
cat $O/rccl_world1_timing.json | grep dp_mode
cd /tmp; export TMPDIR=/tmp; rm -rf $GRAFT_REPO_ROOT/$O/r5_g_prof2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/r5_g_prof2 -- python $GRAFT_REPO_ROOT/tools/allreduce_step_trace.py --rows 16384 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/$O/r5_g_prof2 -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/$O/r5_g_kernel_stats_allreduce_world1.csv; rm -rf $GRAFT_REPO_ROOT/$O/r5_g_prof2; head -8 $GRAFT_REPO_ROOT/$O/r5_g_kernel_stats_allreduce_world1.csv | cut -c1-170
