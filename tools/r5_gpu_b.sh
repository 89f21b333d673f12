cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_general_mlp_gpu.py -q -m gpu --tb=short --show-capture=no -x 2>&1 | tail -25 > $O/r5_b_pytest.log; tail -25 $O/r5_b_pytest.log
for s in 1 0; do OSA_GMLP_SKINNY=$s timeout 300 python tools/general_mlp_timing.py --shapes 1024x1024:64 512x512x512:64 256x128:64 1024x1024:32 --reps 20 --out $O/r5_b_gm_timing_skinny$s.json 2>&1 | grep -v amdgpu | tail -8; done
cd /tmp; export TMPDIR=/tmp; rm -rf $GRAFT_REPO_ROOT/$O/r5_b_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/r5_b_prof -- python $GRAFT_REPO_ROOT/tools/general_mlp_timing.py --shapes 1024x1024:64 --reps 10 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/$O/r5_b_prof -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/$O/r5_b_kernel_stats_general_1024_B64.csv; rm -rf $GRAFT_REPO_ROOT/$O/r5_b_prof; head -14 $GRAFT_REPO_ROOT/$O/r5_b_kernel_stats_general_1024_B64.csv | cut -c1-150
