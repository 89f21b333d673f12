cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 60 tools/_probe/p_base 1024 64 0 2>&1 | tee $O/r5_skinny_probe3.txt
timeout 900 python -m pytest tests/test_general_mlp_gpu.py tests/test_siblings_gpu.py -q -m gpu --tb=short --show-capture=no -k "general or skinny or oracle or extended" 2>&1 | tail -30 > $O/r5_d_pytest.log; tail -30 $O/r5_d_pytest.log | cut -c1-250
timeout 600 python -m pytest tests/test_rccl_gpu.py -q -m gpu --tb=short --show-capture=no -x 2>&1 | tail -15 > $O/r5_d_pytest_rccl.log; tail -15 $O/r5_d_pytest_rccl.log | cut -c1-250
for b in 1 0; do OSA_GMLP_BIG=$b timeout 300 python tools/general_mlp_timing.py --shapes 1024x1024:64 512x512x512:64 256x128:64 --reps 20 --out $O/r5_d_gm_timing_big$b.json 2>&1 | grep -v amdgpu | tail -4; done
cd /tmp; export TMPDIR=/tmp; rm -rf $GRAFT_REPO_ROOT/$O/r5_d_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/r5_d_prof -- python $GRAFT_REPO_ROOT/tools/general_mlp_timing.py --shapes 1024x1024:64 --reps 10 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/$O/r5_d_prof -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/$O/r5_d_kernel_stats_general_1024_B64.csv; rm -rf $GRAFT_REPO_ROOT/$O/r5_d_prof; head -9 $GRAFT_REPO_ROOT/$O/r5_d_kernel_stats_general_1024_B64.csv | cut -c1-150
cat $GRAFT_REPO_ROOT/$O/rccl_world1_timing.json
