R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_general_mlp_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python tools/general_mlp_timing.py --out gpurun_out/r4_general_mlp_timing.json 2>&1 | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp
for spec in 1024x1024:16384 1024x1024:64; do
rm -rf $R/gpurun_out/r4_prof_gm
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_prof_gm -- python $R/tools/general_mlp_timing.py --shapes $spec --reps 5 > /dev/null 2>&1
f=$(find $R/gpurun_out/r4_prof_gm -name "*kernel_stats.csv" | head -1); echo "== $spec"; head -13 $f | cut -c1-150
done
