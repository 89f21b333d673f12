# GPU: correctness of the persistent pass (vs per-step kernels, vs the reference golden) + headline timing
timeout 300 python -m pytest tests/test_mlp_gpu.py -x -q -m gpu -k "pass or ppolag_update or replicated" 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-variant --steps 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('env-steps/s', d['value'], 'us/step', d['roofline']['us_per_optimiser_step'])"
