timeout 1500 python -m pytest tests/test_fragment_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|assert|^E " | tail -12
