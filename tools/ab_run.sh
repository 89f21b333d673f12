# GPU side of the A/B harness: alternates the libraries given as arguments ("" = default build) three times
for i in 1 2 3; do
  for v in "$@"; do
    [ "$v" = "new" ] && p="" || p=omnisafe_amd/lib/libomnisafe_amd_$v.so
    OSA_LIB_PATH=$p timeout 300 python bench.py --no-cpu-baseline --no-variant --steps 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['us_per_optimiser_step'])"
  done
done
