#!/usr/bin/env python3
"""Times the reference's OWN Python step() in the build container (needs /root/reference; never runs on the GPU box).
Same synthetic inputs as bench.py / SURVEY.md 8d: grid BS layout, all UEs 'slow', mixed sharing, log utility, reward
avg, uniform random actions, reset every 100 steps, rand_episodes=True.  The shim Point.distance is a plain sqrt, i.e.
lighter than real shapely/GEOS: these figures are optimistic for the reference.
Usage: python tests/golden/time_reference.py > profiles/archive/r01_reference_cpu_timing.txt"""
import os
import platform
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, '/root/reference')
import _ref_shims  # noqa: E402

_ref_shims.install()
import logging  # noqa: E402

logging.disable(logging.CRITICAL)
from shapely.geometry import Point  # noqa: E402
from deepcomp.env.entities.map import Map  # noqa: E402
from deepcomp.env.entities.station import Basestation  # noqa: E402
from deepcomp.env.entities.user import User  # noqa: E402
from deepcomp.env.util.movement import RandomWaypoint  # noqa: E402
from deepcomp.env.multi_ue.central import CentralRelNormEnv  # noqa: E402
from deepcomp.env.multi_ue.multi_agent import MultiAgentMobileEnv  # noqa: E402
from deepcomp_amd import scenarios  # noqa: E402


def run(kind, U, B, steps):
    scn = scenarios.grid_map(B, 'mixed').with_ues(num_slow=U)
    m = Map(scn.width, scn.height)
    bs = [Basestation(i, Point(x, y), s) for i, (x, y), s in zip(scn.bs_ids, scn.bs_pos, scn.bs_sharing)]
    ues = [User(s['id'], m, 'random', 'random', RandomWaypoint(m, velocity='slow')) for s in scn.ue_specs]
    cfg = {'episode_length': 100, 'seed': 42, 'map': m, 'bs_list': bs, 'ue_list': ues, 'rand_episodes': True,
           'new_ue_interval': None, 'reward': 'avg', 'max_ues': None, 'ue_arrival': None, 'log_metrics': True,
           'dashboard': False, 'ue_details': False}
    env = (CentralRelNormEnv if kind == 'central' else MultiAgentMobileEnv)(cfg)
    rng = random.Random(7)
    env.reset()
    t0 = time.perf_counter()
    for t in range(steps):
        if t % 100 == 0:
            env.reset()
        a = [rng.randint(0, B) for _ in range(U)] if kind == 'central' else {ue.id: rng.randint(0, B) for ue in ues}
        env.step(a)
    dt = time.perf_counter() - t0
    return steps / dt


if __name__ == '__main__':
    print(f'# reference step() in the build container: {platform.processor() or platform.machine()}, 1 core, Python {platform.python_version()}')
    print('# env_class  UxB  env-steps/s  ms/env-step  pair-steps/s')
    for kind, U, B, steps in [('central', 3, 3, 3000), ('central', 10, 5, 800), ('multi', 32, 10, 200), ('multi', 128, 32, 30)]:
        r = run(kind, U, B, steps)
        print(f'{kind:8s} {U}x{B}  {r:10.1f}  {1e3 / r:8.3f}  {r * U * B:10.0f}')
