# round 3, final measurement set with the final kernels: round check + rocprofv3 / PMC passes of the bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/r3_gpu_round_check.sh
sed -i 's/^for i in 1 2 3; do$/for i in 1; do/' tools/r3_gpu_batch8.sh
bash tools/r3_gpu_batch8.sh
timeout 600 python tools/dp_shapes_timing.py --out gpurun_out/r3_dp_shapes_timing.json 2>&1 | grep -v "^{" | grep -v amdgpu
