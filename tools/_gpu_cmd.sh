export DCOMP_LIB=$GRAFT_REPO_ROOT/deepcomp_amd/csrc/variants/libdcomp_hip_r3f.so
python -m pytest tests/test_parity_gpu.py tests/test_adapters_gpu.py tests/test_rollout_gpu.py -q -m gpu -k "128 or 32 or per_gpu_shares or dense_cells or full_size" 2>&1 | grep -v "no kernel built" | tail -30 > gpurun_out/r3_t7_pytest.log
unset DCOMP_LIB
python tools/ab_lib.py run r2 r3e r3f --rounds 2 --only c5 > gpurun_out/r3_t7_ab.log 2>&1
grep -c "no kernel built" gpurun_out/r3_t7_pytest.log; grep "passed\|failed" gpurun_out/r3_t7_pytest.log; tail -4 gpurun_out/r3_t7_ab.log
