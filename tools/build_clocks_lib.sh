#!/bin/bash
# Builds omnisafe_amd/lib/libomnisafe_amd_clocks.so: the product library with the pass kernel's s_memtime
# phase clocks compiled in (-DOSA_PASS_CLOCKS).  Use: OSA_LIB_PATH=omnisafe_amd/lib/libomnisafe_amd_clocks.so
# python tools/phase_clocks.py   (or tools/dp_timing.py)
set -e
cd "$(dirname "$0")/.."
python -m omnisafe_amd.build >/dev/null
L=omnisafe_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DOSA_PASS_CLOCKS -c omnisafe_amd/csrc/ppo_pass_kernel.hip -o $L/ppo_pass_kernel_clocks.hip.o -Wall -Wno-unused-function
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libomnisafe_amd_clocks.so $L/buffer_kernels.hip.o $L/mlp_kernels.hip.o $L/rollout_kernels.hip.o $L/ppo_pass_kernel_clocks.hip.o
echo $L/libomnisafe_amd_clocks.so
