cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/r4_variant_trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/r4_variant_trace -o run -- python $R/tools/variant_timeline.py --run > $O/r4_variant_trace.log 2>&1
f=$(find $O/r4_variant_trace -name "*kernel_trace.csv" | head -1)
python $R/tools/variant_timeline.py --analyse $f --out $O/r4_variant_timeline.json | tail -40
rm -rf $O/r4_variant_trace
