# round 3: LDS / wait counters of the pass kernel (one PMC pass over the bench with 2 update passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r3_pmc_lds
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_UNALIGNED_STALL --output-format csv -d $R/gpurun_out/r3_pmc_lds -- python $R/bench.py --steps 1 --warmup 2 --update-iters 2 --no-cpu-baseline --no-variant > $R/gpurun_out/r3_pmc_lds.log 2>&1
tail -2 $R/gpurun_out/r3_pmc_lds.log
python - <<'PY'
import csv, glob, collections, os
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for p in glob.glob(R + '/gpurun_out/r3_pmc_lds/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if not k.startswith('osa_ppo_pass'):
            continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
            n[k] += 1
for k, v in agg.items():
    print(k, 'launches', n[k])
    for c, x in sorted(v.items()):
        print('   ', c, x / max(n[k], 1))
PY
