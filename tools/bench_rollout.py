#!/usr/bin/env python3
"""GPU box: fused-rollout timing of one shape (bench.py's measure_rollout, stand-alone).
usage: python tools/bench_rollout.py E U B kind T [steps]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deepcomp_amd import scenarios
from deepcomp_amd.entities import build_from_scenario
from deepcomp_amd.env import BatchedMobileEnv

E, U, B, kind, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 4000
r = bench.measure_rollout(torch, BatchedMobileEnv, scenarios, build_from_scenario, torch.device('cuda', 0), E, U, B, kind, T=T, steps=steps,
                          launches_too='--launches' in sys.argv)
print(json.dumps({k: (round(v, 6) if isinstance(v, float) else v) for k, v in r.items()}))
