# round 3, GPU call 1: the GPU suite on the advice fixes + host-env bridge, and ONE FULL (40-pass, nothing
# extrapolated) epoch of the unmodified reference at the benchmark shape on this box's host cores
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r3_b1_pytest.log
timeout 900 python oracle/ref_cpu_baseline.py --sample-iters 40 --threads 16 > gpurun_out/r3_ref_full_epoch_config2.json 2> gpurun_out/r3_ref_full_epoch.err; cat gpurun_out/r3_ref_full_epoch_config2.json
nproc
