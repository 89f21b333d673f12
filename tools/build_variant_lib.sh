#!/bin/bash
# Builds omnisafe_amd/lib/libomnisafe_amd_<suffix>.so: the product library with ONE kernel source recompiled
# with extra flags, e.g.
#     tools/build_variant_lib.sh clocks ppo_pass_kernel.hip -DOSA_PASS_CLOCKS
#     tools/build_variant_lib.sh wclocks wide_pass_kernel.hip -DOSA_WIDE_CLOCKS
# Use it through OSA_LIB_PATH=omnisafe_amd/lib/libomnisafe_amd_<suffix>.so (tools only; the product loads the
# default build and checks its source digest).
set -e
cd "$(dirname "$0")/.."
SUF=$1; SRC=$2; shift; shift
python -m omnisafe_amd.build >/dev/null
L=omnisafe_amd/lib
EXTRA=""
[ "$SRC" = "ppo_pass_kernel.hip" -o "$SRC" = "part_grad_kernel.hip" -o "$SRC" = "p2p_pass_kernel.hip" ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $EXTRA "$@" -c omnisafe_amd/csrc/$SRC -o $L/${SRC}_$SUF.o -Wall -Wno-unused-function
OBJS=""
for o in $L/*.hip.o; do
  [ "$(basename $o)" = "$SRC.o" ] && continue
  OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libomnisafe_amd_$SUF.so $OBJS $L/${SRC}_$SUF.o
echo $L/libomnisafe_amd_$SUF.so
