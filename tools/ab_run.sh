# GPU side of tools/ab_pass.sh: alternates base / new three times
for i in 1 2 3; do
  for v in omnisafe_amd/lib/libomnisafe_amd_base.so ""; do
    OSA_LIB_PATH=$v timeout 300 python bench.py --no-cpu-baseline --no-variant --steps 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${v:-new }', d['value'], d['roofline']['us_per_optimiser_step'])"
  done
done
