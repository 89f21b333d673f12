cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_general_mlp_gpu.py tests/test_siblings_gpu.py -q -m gpu --tb=short --show-capture=no -k "general or skinny or oracle or extended" 2>&1 | tail -8 > $O/r5_e_pytest.log; tail -8 $O/r5_e_pytest.log | cut -c1-250
for d in 0 12000 22000 48000; do echo "dynlds $d"; OSA_GS_WGRAD_DYNLDS=$d timeout 300 python tools/general_mlp_timing.py --shapes 1024x1024:64 512x512x512:64 --reps 20 2>&1 | grep -v amdgpu | tail -2; done
cd /tmp; export TMPDIR=/tmp
for d in 12000 22000; do rm -rf $GRAFT_REPO_ROOT/$O/r5_e_prof
OSA_GS_WGRAD_DYNLDS=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/r5_e_prof -- python $GRAFT_REPO_ROOT/tools/general_mlp_timing.py --shapes 1024x1024:64 --reps 10 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/$O/r5_e_prof -name "*kernel_stats.csv" | head -1); echo "dynlds $d"; head -6 $f | cut -c1-150; done; rm -rf $GRAFT_REPO_ROOT/$O/r5_e_prof
