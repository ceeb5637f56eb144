#!/usr/bin/env python3
"""GPU box: the heuristic policy kernel (dcomp_heuristic_actions) alone and in the loop with dcomp_step.
usage: python tools/bench_policy.py [--envs 65536 --ues 32 --bs 10 --kind multi]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcomp_amd import agents, scenarios                        # noqa: E402
from deepcomp_amd.entities import build_from_scenario             # noqa: E402
from deepcomp_amd.env import BatchedMobileEnv                     # noqa: E402


def timed(fn, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=65536)
    ap.add_argument('--ues', type=int, default=32)
    ap.add_argument('--bs', type=int, default=10)
    ap.add_argument('--kind', default='multi')
    a = ap.parse_args()
    E, U, B = a.envs, a.ues, a.bs
    scn = scenarios.grid_map(B, 'mixed').with_ues(num_slow=U)
    m, bs, ues = build_from_scenario(scn)
    env = BatchedMobileEnv(m, bs, ues, a.kind, num_envs=E, seed=42, rng='philox', rand_episodes=True, episode_length=100)
    env.reset()
    act = torch.zeros((E, U), dtype=torch.uint8, device='cuda')
    ags = {'3gpp': agents.Heuristic3GPP(), 'fullcomp': agents.FullCoMP(), 'dynamic(0.3)': agents.DynamicSelection(0.3),
           'cluster(3)': agents.StaticClustering(3, bs, seed=1, device='cuda')}
    read = E * U * 2 * B * 4 + E * U                                 # connected | dr floats read + the action byte
    for _ in range(30):
        env.step(act)
    step_ms = timed(lambda: env.step(act), 300)
    print(f'{E} x {U} x {B} {a.kind}: dcomp_step alone {step_ms:.4f} ms')
    for name, ag in ags.items():
        for _ in range(20):
            ag.act(env, out=act)
        k = timed(lambda: ag.act(env, out=act), 300)
        views = env.obs_views() if a.kind == 'multi' else None
        t = timed(lambda: ag(views), 30) if views is not None else float('nan')

        def loop():
            ag.act(env, out=act)
            env.step(act)
        env.reset()
        for _ in range(20):
            loop()
        env.reset()
        lp = timed(loop, 90)
        def fused():
            env.step(ag.act(env))                    # after the first call: the step kernel wrote next_action itself
        env.reset()
        for _ in range(20):
            fused()
        env.reset()
        fl = timed(fused, 90) if env.next_action is not None else float('nan')
        if env.next_action is not None and env.fused_rollout:
            env.reset()
            env.rollout_policy(99)
            env.reset()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                env.rollout_policy(99)
                env.reset()
            e1.record()
            torch.cuda.synchronize()
            per = e0.elapsed_time(e1) / 1000
            print(f'  {name:13s} closed loop inside the fused rollout (policy_loop, 99 steps + reset per launch pair) {per * 1e3:.2f} us/step = {E / per * 1e3:.3e} env-steps/s')
        env.set_policy(None)
        print(f'  {name:13s} in-step policy (dcomp_set_policy) + step {fl:.4f} ms = {E / fl * 1e3:.3e} env-steps/s')
        print(f'  {name:13s} kernel {k:.4f} ms ({read / k / 1e6:.0f} GB/s of connected|dr)   tensor-expression form {t:.3f} ms   '
              f'policy + step {lp:.4f} ms = {E / lp * 1e3:.3e} env-steps/s')
    env.check()


if __name__ == '__main__':
    main()
