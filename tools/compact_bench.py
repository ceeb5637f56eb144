#!/usr/bin/env python3
"""GPU box: the step with rows, the step + pack_fragment, and the step that writes the compact record itself (dcomp_out.obs_compact).
usage: python tools/compact_bench.py [E U B]...   (default: the BASELINE shapes)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcomp_amd import scenarios                                # noqa: E402
from deepcomp_amd.entities import build_from_scenario             # noqa: E402
from deepcomp_amd.env import BatchedMobileEnv                     # noqa: E402
from deepcomp_amd.fragment import FragmentCodec                   # noqa: E402


def timed(fn, n, warm):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    shapes = [(65536, 32, 10), (262144, 32, 10), (4096, 128, 32), (8192, 128, 32), (32768, 128, 32)]
    if len(sys.argv) > 3:
        v = [int(x) for x in sys.argv[1:]]
        shapes = [tuple(v[i:i + 3]) for i in range(0, len(v), 3)]
    for E, U, B in shapes:
        m, bs, ues = build_from_scenario(scenarios.grid_map(B, 'mixed').with_ues(num_slow=U // 2, num_fast=U - U // 2))
        env = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=42, rng='philox', rand_episodes=True, episode_length=100000, log_metrics=True)
        env.reset()
        codec = FragmentCodec(U, B)
        g = torch.Generator(device='cuda').manual_seed(1)
        act = torch.randint(0, B + 1, (E, U), generator=g, device='cuda', dtype=torch.uint8)
        packed = torch.empty((E, codec.words), dtype=torch.int32, device='cuda')
        rew = torch.empty_like(env.reward)
        n, warm = (300, 300) if E * U * B < 5e7 else (100, 60)
        rows = timed(lambda: env.step(act), n, warm)
        pk = timed(lambda: codec.pack(env.obs, out=packed), n, 20)

        def both():
            env.step(act)
            codec.pack(env.obs, out=packed)
        sp = timed(both, n, 20)
        cp = timed(lambda: env.step_compact(act, packed, rew), n, warm)
        rows2 = timed(lambda: env.step(act), n, warm)
        row_b, rec_b = U * (4 * B + 1) * 4, codec.words * 4
        print(f'{E} x {U} x {B} ({env.step_kernel_name}): rows {rows:.1f} us (again {rows2:.1f})   pack alone {pk:.1f}   step + pack {sp:.1f}   '
              f'compact step {cp:.1f} us = {E / cp:.1f} env-steps/us   [{row_b} -> {rec_b} B of observation per env-step]', flush=True)
        env.check()
        del env, packed, act, rew
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
