R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
for i in 1 2 3 4; do
  timeout 600 python -m pytest tests -q -m gpu --tb=long -p no:cacheprovider > /tmp/full_$i.log 2>&1
  tail -2 /tmp/full_$i.log
  if grep -q "failed" /tmp/full_$i.log; then cp /tmp/full_$i.log gpurun_out/r3_flaky_full_$i.log; fi
done
