export DCOMP_LIB=$GRAFT_REPO_ROOT/deepcomp_amd/csrc/variants/libdcomp_hip_w3.so
python -m pytest tests/test_parity_gpu.py tests/test_adapters_gpu.py -q -m gpu -k "128 or per_gpu_shares or dense_cells" 2>&1 | grep -v "no kernel built" | tail -8 > gpurun_out/r3_t10_pytest.log
unset DCOMP_LIB
python tools/ab_lib.py run r3f w3 --rounds 2 --only c5 > gpurun_out/r3_t10_ab.log 2>&1
grep "passed\|failed" gpurun_out/r3_t10_pytest.log; tail -3 gpurun_out/r3_t10_ab.log
