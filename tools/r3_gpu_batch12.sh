# round 3: kernel-trace timeline of the large-batch variant's epoch
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
rm -rf $R/gpurun_out/r3_variant_trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3_variant_trace -- python $R/tools/variant_timeline.py --run 2>&1 | grep "^{"
f=$(find $R/gpurun_out/r3_variant_trace -name "*kernel_trace.csv" | head -1)
echo trace: $f
python $R/tools/variant_timeline.py --analyse $f --out $R/gpurun_out/r3_variant_timeline.json | head -400
rm -rf $R/gpurun_out/r3_variant_trace
