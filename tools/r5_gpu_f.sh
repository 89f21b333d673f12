cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_general_mlp_gpu.py tests/test_siblings_gpu.py tests/test_dp_golden_gpu.py -q -m gpu --tb=short --show-capture=no -k "general or skinny or oracle or extended" 2>&1 | tail -12 > $O/r5_f_pytest.log; tail -12 $O/r5_f_pytest.log | cut -c1-250
for f in 1 0; do echo "top fuse $f"; OSA_GMLP_TOP_FUSE=$f timeout 300 python tools/general_mlp_timing.py --shapes 1024x1024:64 512x512x512:64 256x128:64 --reps 20 --out $O/r5_f_gm_timing_fuse$f.json 2>&1 | grep -v amdgpu | tail -3; done
timeout 300 python tools/allreduce_step_trace.py --out $O/r5_allreduce_step.json 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp; rm -rf $GRAFT_REPO_ROOT/$O/r5_f_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/r5_f_prof -- python $GRAFT_REPO_ROOT/tools/general_mlp_timing.py --shapes 1024x1024:64 --reps 10 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/$O/r5_f_prof -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/$O/r5_f_kernel_stats_general_1024_B64.csv; rm -rf $GRAFT_REPO_ROOT/$O/r5_f_prof; head -8 $GRAFT_REPO_ROOT/$O/r5_f_kernel_stats_general_1024_B64.csv | cut -c1-150
rm -rf $GRAFT_REPO_ROOT/$O/r5_f_prof2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/r5_f_prof2 -- python $GRAFT_REPO_ROOT/tools/allreduce_step_trace.py --rows 16384 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/$O/r5_f_prof2 -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/$O/r5_f_kernel_stats_allreduce_world1.csv; rm -rf $GRAFT_REPO_ROOT/$O/r5_f_prof2; head -8 $GRAFT_REPO_ROOT/$O/r5_f_kernel_stats_allreduce_world1.csv | cut -c1-170
