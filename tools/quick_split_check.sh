# quick GPU check of the wide-observation passes: parity tests, then us per step (and phase clocks with the
# -DOSA_SPLIT_CLOCKS build of tools/build_variant_lib.sh)
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_mlp_gpu.py tests/test_config_shapes_gpu.py -q -k "persistent_pass_equals or config4 or first_order" 2>&1 | tail -5
[ -f omnisafe_amd/lib/libomnisafe_amd_sclocks.so ] && OSA_LIB_PATH=omnisafe_amd/lib/libomnisafe_amd_sclocks.so timeout 300 python tools/wide_pass_timing.py 65536 2>&1 | grep -B7 "^376/17: persistent-wide-split (local)"
timeout 300 python tools/wide_pass_timing.py 65536 2>&1 | grep "split"
