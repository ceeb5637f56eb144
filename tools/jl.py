#!/usr/bin/env python3
"""Print selected fields of bench.py's JSON line (stdin): tools/jl.py <label>"""
import json
import sys
lines = [l for l in sys.stdin.read().splitlines() if l.startswith('{')]
if not lines:
    print(sys.argv[1:], 'NO JSON LINE')
    sys.exit(0)
j = json.loads(lines[-1])
print(' '.join(sys.argv[1:]), f"value={j['value']:.4g} ms_per_step={j['ms_per_step']:.5f} kernel_ms={j['roofline']['kernel_ms']:.5f} frac={j['roofline']['frac']:.3f}")
