cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_mlp_gpu.py tests/test_rccl_gpu.py tests/test_dp_golden_gpu.py tests/test_general_mlp_gpu.py tests/test_rollout_gpu.py -x -q -m gpu 2>&1 | tail -6
timeout 600 python bench.py --steps 4 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); v=j['throughput_variant']; print(j['value'], v['value'], v['ms_per_step'], v['update_path'], v['roofline']['us_per_optimiser_step'])"
OSA_UPDATE_GRAPH_WHOLE=0 timeout 600 python bench.py --steps 4 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); v=j['throughput_variant']; print('whole=0', j['value'], v['value'], v['ms_per_step'], v['update_path'], v['roofline']['us_per_optimiser_step'])"
