#!/usr/bin/env python3
"""Steps/s of the drop-in single-env classes (num_envs = 1, reference-shaped Python returns, one host round trip per
step) -- the mode an unmodified RLlib worker uses.  Compare BASELINE.md: reference env 69.9 steps/s at 32 x 10."""
import sys
import time

sys.path.insert(0, '.')
from deepcomp_amd import scenarios
from deepcomp_amd.entities import make_env_config
from deepcomp_amd.env import CentralRelNormEnv, MultiAgentMobileEnv
import random

for name, cls, scn in [('multi 32x10', MultiAgentMobileEnv, scenarios.grid_map(10, 'mixed').with_ues(num_slow=32)),
                       ('central 10x5', CentralRelNormEnv, scenarios.grid_map(5, 'mixed').with_ues(num_slow=10)),
                       ('central 3x3', CentralRelNormEnv, scenarios.medium_map('mixed').with_ues(num_slow=3))]:
    env = cls(make_env_config(scn, seed=42))
    rng = random.Random(1)
    U, B = env.num_ue, env.num_bs
    env.reset()
    n = 300
    acts = [({ue.id: rng.randint(0, B) for ue in env.ue_list} if cls is MultiAgentMobileEnv else [rng.randint(0, B) for _ in range(U)])
            for _ in range(n)]
    for a in acts[:20]:
        env.step(a)
    t0 = time.perf_counter()
    for i, a in enumerate(acts):
        if i % 100 == 0:
            env.reset()
        env.step(a)
    dt = time.perf_counter() - t0
    print(f'{name}: {n / dt:.0f} env-steps/s ({dt / n * 1e3:.2f} ms per step, E = 1, PCIe round trip included)')
