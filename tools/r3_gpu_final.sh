# round 3, final measurement set with the final kernels: GPU suite + bench + 2-rank check + configs + GAE + pass timings
# (tools/r3_gpu_round_check.sh), rocprofv3 stats and PMC passes of the bench (tools/r3_gpu_batch8.sh, one A/B loop), the
# data-parallel shapes, the race hunt, the variant's timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/r3_gpu_round_check.sh
sed -i 's/^for i in 1 2 3; do$/for i in 1; do/' tools/r3_gpu_batch8.sh
bash tools/r3_gpu_batch8.sh
cd $R
timeout 600 python tools/dp_shapes_timing.py --out gpurun_out/r3_dp_shapes_timing.json 2>&1 | grep -v "^{" | grep -v amdgpu
timeout 600 python tools/dp_stress.py --iters 3000 --out gpurun_out/r3_dp_stress.json 2>&1 | grep -v amdgpu | cut -c1-300
bash tools/r3_gpu_batch12.sh 2>&1 | head -12
timeout 300 python tools/large_batch_step_timing.py --out gpurun_out/r3_large_batch_step.json 2>&1 | grep -v amdgpu | tail -8
