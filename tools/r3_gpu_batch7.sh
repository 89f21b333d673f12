set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_buffer_gpu.py -q -x -m gpu -k chained 2>&1 | tail -3
SH="16,65536 64,32768 128,16384 128,65536 256,4096 256,65536 1024,64 1024,1024 4096,256 4096,4096 5000,4"
for v in "" gcnw4 gcnw12; do
  echo "== variant ${v:-default}"
  [ -n "$v" ] && export OSA_LIB_PATH=omnisafe_amd/lib/libomnisafe_amd_$v.so || unset OSA_LIB_PATH
  timeout 600 python tools/gae_bandwidth.py --only-gae --out gpurun_out/r3_gae_bw_${v:-default} --shapes $SH 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['T'], r['N'], 'seq', r['gae_sequential_us'], 'tiled', r['gae_tiled_us'], 'chained', r['gae_chained_us'], r['gae_chained_GBps'])"
done
