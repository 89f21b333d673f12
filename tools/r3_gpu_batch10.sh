# round 3: the three tests that failed in the final full-suite run, with tracebacks; the hand-off protocol probe
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_buffer_gpu.py::test_cabi_error_codes "tests/test_learning_gpu.py::test_sibling_learning_curve_within_one_sigma_of_reference" "tests/test_mlp_gpu.py::test_wide_split_data_parallel_pass_equals_allreduce_semantics" -q -m gpu 2>&1 | tail -150 > gpurun_out/r3_b10_pytest.log
tail -5 gpurun_out/r3_b10_pytest.log
timeout 300 tools/bin/handoff_probe 2000 20000 > gpurun_out/r3_handoff_probe.json 2> gpurun_out/r3_handoff_probe.err
tail -c 300 gpurun_out/r3_handoff_probe.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r3_handoff_probe.json'))
for r in d['rows']:
    print(r['placement'][:10], r['W'], r['skew_mask'], r['protocol'][:30].ljust(30), r['us_per_iter'], r['clocks_per_iter'], r['clocks_in_handoff'], r['timeout'], r['wrong_sums'], r['xcc_mask'])
PY
