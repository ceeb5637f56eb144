export DCOMP_LIB=$GRAFT_REPO_ROOT/deepcomp_amd/csrc/variants/libdcomp_hip_wb128.so
python -m pytest tests/test_parity_gpu.py tests/test_adapters_gpu.py -q -m gpu -k "128 or 256 or 200 or 100 or 64 or per_gpu_shares or dense_cells" 2>&1 | grep -v "no kernel built" | tail -8 > gpurun_out/r3_t15_pytest.log
unset DCOMP_LIB
python tools/ab_lib.py run wb256 wb128 --rounds 3 --only c5,c5big > gpurun_out/r3_t15_ab.log 2>&1
grep "passed\|failed" gpurun_out/r3_t15_pytest.log; tail -4 gpurun_out/r3_t15_ab.log
