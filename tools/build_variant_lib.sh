#!/bin/bash
# Builds omnisafe_amd/lib/libomnisafe_amd_<suffix>.so: the product library with ppo_pass_kernel.hip compiled
# with extra flags (e.g. `clocks -DOSA_PASS_CLOCKS`, `ru2 -DOSA_DP_RU=2`).  Use it through
# OSA_LIB_PATH=omnisafe_amd/lib/libomnisafe_amd_<suffix>.so (tools only; the product loads the default build).
set -e
cd "$(dirname "$0")/.."
SUF=$1; shift
python -m omnisafe_amd.build >/dev/null
L=omnisafe_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -c omnisafe_amd/csrc/ppo_pass_kernel.hip -o $L/ppo_pass_kernel_$SUF.hip.o -Wall -Wno-unused-function
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libomnisafe_amd_$SUF.so $L/buffer_kernels.hip.o $L/mlp_kernels.hip.o $L/rollout_kernels.hip.o $L/ppo_pass_kernel_$SUF.hip.o
echo $L/libomnisafe_amd_$SUF.so
