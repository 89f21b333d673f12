set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
L=omnisafe_amd/lib
OSA_LIB_PATH=$L/libomnisafe_amd_sclocks.so timeout 300 python tools/split_dp_clocks.py 1 8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_split_dp_clocks.log
for v in ru1 ru4 ru8; do
  echo "== $v"
  OSA_LIB_PATH=$L/libomnisafe_amd_$v.so timeout 300 python tools/split_dp_clocks.py 4 8 2>&1 | grep "^W=" | tee -a gpurun_out/r3_split_dp_ru.log
done
