date
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|error" | tail -15
date
python tools/compact_bench.py 2>&1 | grep " x "
