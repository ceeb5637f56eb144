#!/usr/bin/env python3
"""Generic kernel (more than 32 stations / 256 UE slots): rollout() as ONE launch per stretch (big_kernel<..., ROLL>, round 6) against one
launch per step inside the same call (DCOMP_NO_FUSED_BIG=1).  HIP events around `reps` rollouts of T steps, every step's outputs written.
    python tools/big_rollout_rate.py [--T 50]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deepcomp_amd import scenarios
from deepcomp_amd.entities import build_from_scenario
from deepcomp_amd.env import BatchedMobileEnv

SHAPES = [('multi', 64, 32, 64), ('multi', 512, 32, 64), ('multi', 2048, 32, 64), ('multi', 8192, 32, 64), ('central', 256, 10, 40), ('central', 4096, 10, 40),
          ('central', 65536, 10, 40), ('multi', 16, 300, 12), ('multi', 256, 300, 12)]


def rate(kind, E, U, B, T, fused, every, reps):
    if fused:
        os.environ.pop('DCOMP_NO_FUSED_BIG', None)
    else:
        os.environ['DCOMP_NO_FUSED_BIG'] = '1'
    m, bs, ues = build_from_scenario(scenarios.grid_map(B, 'mixed').with_ues(num_slow=U - U // 4, num_fast=U // 4))
    env = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=3, rng='philox', rand_episodes=True, episode_length=10 ** 6)
    assert env.rollout_is_fused(T) == fused
    env.reset()
    g = torch.Generator(device='cuda').manual_seed(1)
    acts = torch.randint(0, B + 1, (T, E, U), generator=g, device='cuda', dtype=torch.uint8)
    out = None
    if every:
        out = {'obs': torch.empty((T,) + tuple(env.obs.shape), device='cuda'), 'reward': torch.empty((T,) + tuple(env.reward.shape), device='cuda')}
    for _ in range(3):
        env.rollout(acts, out=out)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        env.rollout(acts, out=out)
    b.record()
    torch.cuda.synchronize()
    env.check()
    return a.elapsed_time(b) * 1e3 / (reps * T)          # us per step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--T', type=int, default=50)
    a = ap.parse_args()
    print(f'{"shape":>28} {"outputs":>10} {"per-step launches":>18} {"fused":>10} {"ratio":>7}   (us per step, T = {a.T})')
    for kind, E, U, B in SHAPES:
        for every in (True, False):
            if every and a.T * E * U * (4 * B + 1) * 4 > 24e9:
                continue
            reps = max(2, min(40, int(2e9 / (a.T * E * U * B * 40))))
            s = rate(kind, E, U, B, a.T, False, every, reps)
            f = rate(kind, E, U, B, a.T, True, every, reps)
            print(f'{f"{E} x {U} x {B} {kind}":>28} {"every step" if every else "last step":>10} {s:18.2f} {f:10.2f} {s / f:7.2f}', flush=True)


if __name__ == '__main__':
    main()
