R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in $(seq 1 25); do
  timeout 300 python -m pytest tests/test_mlp_gpu.py -q -m gpu -x --tb=long -k "wide_split_data_parallel" > /tmp/flaky_$i.log 2>&1 || { echo "FAILED at iteration $i"; grep -E "^E |Error|assert" /tmp/flaky_$i.log | head -40; break; }
done
echo "loop done $i"
