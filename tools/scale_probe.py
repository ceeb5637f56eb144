#!/usr/bin/env python3
"""GPU box: kernel time per step of one shape over batch sizes -- separates the per-launch constant (kernel-to-kernel gap, ramp,
tail) from the per-env cost.  usage: python tools/scale_probe.py U B kind E1,E2,..."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deepcomp_amd import scenarios
from deepcomp_amd.entities import build_from_scenario
from deepcomp_amd.env import BatchedMobileEnv

U, B, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rows = []
for E in [int(x) for x in sys.argv[4].split(',')]:
    r = bench.measure_steps(torch, BatchedMobileEnv, scenarios, build_from_scenario, torch.device('cuda', 0), E, U, B, kind)
    rows.append((E, r['kernel_ms'] * 1e3, r['frac_of_hbm_peak'], r['lanes_per_env']))
    print(f'E={E:8d}  {r["kernel_ms"] * 1e3:8.2f} us  {100 * r["frac_of_hbm_peak"]:5.1f} % of HBM peak  lanes/env {r["lanes_per_env"]}', flush=True)
if len(rows) >= 2:
    (e0, t0, _, _), (e1, t1, _, _) = rows[0], rows[-1]
    slope = (t1 - t0) / (e1 - e0)
    print(f'per-env slope {slope * 1e3:.3f} us per 1000 envs; intercept {t0 - slope * e0:.2f} us')
