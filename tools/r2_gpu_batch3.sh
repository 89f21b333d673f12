#!/bin/bash
# Round-2 GPU batch: full GPU suite, GAE bandwidth curve, learning triplets, SQ (MFMA-busy) PMC pass.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -rfE --timeout 900 > gpurun_out/r2_gpu_tests_3.log 2>&1
grep -E "passed|failed|error" gpurun_out/r2_gpu_tests_3.log | tail -3
grep -E "^(FAILED|ERROR)" gpurun_out/r2_gpu_tests_3.log | head -30
python tools/gae_bandwidth.py > gpurun_out/r2_gae_bw.log 2>&1; tail -3 gpurun_out/r2_gae_bw.log
python tools/learning_triplets.py train PPOLag TRPOLag CPO --seeds 60 > gpurun_out/r2_triplets_train.log 2>&1; tail -3 gpurun_out/r2_triplets_train.log
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/r2_counters.txt 2>&1
C=$(grep -o -E "SQ_(INSTS_VALU_MFMA_MOPS_F32|VALU_MFMA_BUSY_CYCLES|BUSY_CYCLES|WAVE_CYCLES|INSTS_VALU|ACTIVE_INST_VALU|INSTS_MFMA|WAIT_INST_ANY)\b" $R/gpurun_out/r2_counters.txt | sort -u | head -8 | tr '\n' ' ')
echo "SQ counters: $C"
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_sq -- python $R/bench.py --steps 1 --warmup 1 --update-iters 2 --no-cpu-baseline --no-variant > $R/gpurun_out/r2_pmc_sq.log 2>&1
tail -2 $R/gpurun_out/r2_pmc_sq.log | cut -c1-300
find $R/gpurun_out/pmc_sq -name "*counter_collection.csv" | head -3
