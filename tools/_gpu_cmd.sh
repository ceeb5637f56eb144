export DCOMP_LIB=$GRAFT_REPO_ROOT/deepcomp_amd/csrc/variants/libdcomp_hip_r3b.so
python -X faulthandler -m pytest tests/test_rollout_gpu.py tests/test_parity_gpu.py tests/test_adapters_gpu.py -v -m gpu -k "rollout or full_size or per_gpu_shares or golden_env_stack or soak or closed_loop or in_step_policy or movement_parameters" > gpurun_out/r3_t4_pytest.log 2>&1
grep -n "PASSED\|FAILED\|Fatal\|fault\|Memory\|abort" gpurun_out/r3_t4_pytest.log | grep -v "no kernel" | tail -60
