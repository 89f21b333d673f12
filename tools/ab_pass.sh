#!/bin/bash
# A/B of the working-tree pass kernel against a git revision's (default HEAD) ON THE SAME BOX (box-to-box
# noise is ~1 %): builds libomnisafe_amd_base.so from `git show REV:omnisafe_amd/csrc/ppo_pass_kernel.hip`,
# then prints the gpurun command line.  Run here; execute the printed command through gpurun.
set -e
cd "$(dirname "$0")/.."
REV=${1:-HEAD}
python -m omnisafe_amd.build >/dev/null
L=omnisafe_amd/lib
git show $REV:omnisafe_amd/csrc/ppo_pass_kernel.hip > omnisafe_amd/csrc/_ab_base.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -c omnisafe_amd/csrc/_ab_base.hip -o $L/_ab_base.o -Wall -Wno-unused-function
rm -f omnisafe_amd/csrc/_ab_base.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libomnisafe_amd_base.so $L/buffer_kernels.hip.o $L/mlp_kernels.hip.o $L/rollout_kernels.hip.o $L/_ab_base.o
echo "built $L/libomnisafe_amd_base.so from $REV"
