cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_trust_region_gpu.py -x -q -m gpu 2>&1 | tail -8
for f in 1 0; do
OSA_FVP_FAST=$f timeout 300 python bench.py --algo TRPOLag --batch-size 128 --update-iters 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fast=$f', j['value'], j['ms_per_step'], j['roofline_fvp']['us_per_launch'], j['roofline_fvp']['frac'], j['roofline_fvp']['kernel'])"
done
OSA_FVP_FAST=1 timeout 300 python bench.py --algo CPO --batch-size 128 --update-iters 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('CPO', j['value'], j['ms_per_step'], j['roofline_fvp']['us_per_launch'], j['roofline_fvp']['frac'])"
