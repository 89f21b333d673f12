# round 3: PPOSaute learning test, tracebacks only, current library and the A/B base library
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_buffer_gpu.py::test_cabi_error_codes "tests/test_learning_gpu.py::test_sibling_learning_curve_within_one_sigma_of_reference[PPOSaute]" "tests/test_learning_gpu.py::test_sibling_learning_curve_within_one_sigma_of_reference[TRPOSaute]" -q -m gpu --tb=short --show-capture=no 2>&1 | tail -60 > gpurun_out/r3_b11_pytest.log
cat gpurun_out/r3_b11_pytest.log
ls -la tools/bin/libomnisafe_amd_base.so
OSA_LIB_PATH=$R/tools/bin/libomnisafe_amd_base.so timeout 900 python -m pytest "tests/test_learning_gpu.py::test_sibling_learning_curve_within_one_sigma_of_reference[PPOSaute]" -q -m gpu --tb=short --show-capture=no 2>&1 | tail -30
