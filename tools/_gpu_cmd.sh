mkdir -p gpurun_out
python -m pytest tests/test_fragment_gpu.py tests/test_handoff_gpu.py -m gpu -q 2>&1 | tail -8
