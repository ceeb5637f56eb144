mkdir -p gpurun_out
python -m pytest tests/test_bigb_gpu.py tests/test_threshold_gpu.py tests/test_fragment_gpu.py tests/test_adapters_gpu.py -q -m gpu -x > gpurun_out/x4_tests.txt 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/x4_tests.txt | tail -5
