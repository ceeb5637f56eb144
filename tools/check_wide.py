#!/usr/bin/env python3
"""GPU box: the oracle-parity test of tests/test_parity_gpu.py on the shapes that run the wide kernel (development loop)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import test_parity_gpu as t

SHAPES = [('multi', 128, 32, 8, 'avg'), ('multi', 128, 29, 6, 'sum'), ('multi', 100, 32, 5, 'min'), ('multi', 64, 25, 8, 'avg'),
          ('multi', 200, 26, 3, 'avg'), ('central', 128, 32, 4, 'min'), ('central', 256, 31, 2, 'sum'), ('central', 128, 32, 12, 'avg'),
          ('multi', 128, 32, 4, 'avg', 'proportional-fair'), ('multi', 64, 26, 6, 'sum', 'rate-fair'), ('multi', 128, 22, 6, 'avg'),
          ('central', 70, 21, 5, 'sum'), ('multi', 64, 24, 8, 'min', 'rate-fair'), ('multi', 128, 28, 5, 'avg'), ('multi', 90, 28, 6, 'sum', 'proportional-fair'),
          ('multi', 256, 28, 2, 'min'), ('multi', 200, 24, 3, 'avg', 'rate-fair'), ('multi', 32, 10, 64, 'min', 'resource-fair')]
built = os.environ.get('DCOMP_BUILD_B')
ok = 0
for s in SHAPES:
    if built and str(s[2]) not in built.split(','):
        continue
    t.test_oracle_parity_philox(torch, s)
    ok += 1
    print('ok', s, flush=True)
print(f'{ok} shapes match the oracle')
