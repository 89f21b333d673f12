# round 3: the whole GPU suite three times in a row (flakiness check after the dz1 fix), failures with tracebacks
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
: > gpurun_out/r3_flaky4.log
for i in 1 2 3; do
  timeout 900 python -m pytest tests -q -m gpu --tb=short --show-capture=no -p no:cacheprovider 2>&1 | tail -25 >> gpurun_out/r3_flaky4.log
  echo "=== run $i done" >> gpurun_out/r3_flaky4.log
done
grep -n "passed\|failed\|FAILED\|=== run" gpurun_out/r3_flaky4.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
