#!/usr/bin/env python3
"""GPU box: rates of the compact-fragment kernels (dcomp_pack_fragment / dcomp_unpack_fragment) on the step kernels' own
observation tensors.  Bytes moved = rows + compact record (read one, write the other); HIP events over 200 launches."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepcomp_amd import scenarios
from deepcomp_amd.entities import build_from_scenario
from deepcomp_amd.env import BatchedMobileEnv
from deepcomp_amd.fragment import FragmentCodec

for (E, U, B) in ((65536, 32, 10), (4096, 128, 32), (32768, 128, 32), (262144, 32, 10)):
    m, bs, ues = build_from_scenario(scenarios.grid_map(B, 'mixed').with_ues(num_slow=U))
    env = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=3, rng='philox', rand_episodes=True)
    env.reset()
    a = torch.randint(0, B + 1, (E, U), device='cuda', dtype=torch.uint8)
    for t in range(5):
        env.step(a)
    c = FragmentCodec(U, B)
    p = c.pack(env.obs)
    o = c.unpack(p)
    assert torch.equal(o.view(torch.int32), env.obs.view(torch.int32))
    for name, fn in (('pack', lambda: c.pack(env.obs, out=p)), ('unpack', lambda: c.unpack(p, out=o))):
        for _ in range(20):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 200
        byts = env.obs.numel() * 4 + p.numel() * 4
        print(f'{E}x{U}x{B} {name}: {ms * 1e3:.1f} us, {byts / 1e6:.0f} MB moved ({env.obs.numel() * 4 / 1e6:.0f} MB rows <-> {p.numel() * 4 / 1e6:.0f} MB compact), '
              f'{byts / ms / 1e6:.0f} GB/s = {byts / ms / 1e6 / 8000:.3f} of 8 TB/s', flush=True)
    c.check()
    del env
