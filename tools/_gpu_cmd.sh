mkdir -p gpurun_out
python -m pytest tests/test_handoff_gpu.py -m gpu -q -k "compact" 2>&1 | grep -E "passed|failed|Error|assert" | head
