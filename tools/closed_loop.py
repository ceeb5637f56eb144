#!/usr/bin/env python3
"""Policy-in-the-loop sampling rate on one GPU: the batched env feeding a shared-parameter actor (RLlib's default fcnet,
2 x 256 tanh, parameter sharing across UEs as in DD-CoMP, env_setup.py:266-283) that turns the observation tensor into the
next action tensor -- everything stays in HBM, no per-env Python.  The policy is random-init and NOT part of the product;
this only shows where the time goes once the env runs at ~10^8-10^9 env-steps/s.

    python tools/closed_loop.py [--envs 65536] [--steps 200] [--dtype bf16]
"""
import argparse
import sys
import time

sys.path.insert(0, '.')
import torch

from deepcomp_amd import scenarios
from deepcomp_amd.entities import build_from_scenario
from deepcomp_amd.env import BatchedMobileEnv

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=65536)
ap.add_argument('--ues', type=int, default=32)
ap.add_argument('--bs', type=int, default=10)
ap.add_argument('--steps', type=int, default=200)
ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
a = ap.parse_args()

dev = torch.device('cuda', 0)
E, U, B = a.envs, a.ues, a.bs
scn = scenarios.grid_map(B, 'mixed').with_ues(num_slow=U)
m, bs, ues = build_from_scenario(scn)
env = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=42, episode_length=100, rng='philox', rand_episodes=True, device=dev)
dt = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
D = 4 * B + 1
g = torch.Generator(device=dev).manual_seed(0)
W1 = (torch.randn(D, 256, generator=g, device=dev) / D ** 0.5).to(dt)
W2 = (torch.randn(256, 256, generator=g, device=dev) / 16).to(dt)
W3 = (torch.randn(256, B + 1, generator=g, device=dev) / 16).to(dt)


def policy(obs):                                   # obs [E, U, 4B+1] f32 in HBM -> uint8 actions [E, U]
    x = obs.view(E * U, D).to(dt)
    h = torch.tanh(torch.tanh(x @ W1) @ W2)
    logits = (h @ W3).float()
    gumbel = -torch.log(-torch.log(torch.rand_like(logits).clamp_(1e-20, 1.0)))
    return (logits + gumbel).argmax(dim=1).to(torch.uint8).view(E, U)


def run(n, with_policy):
    obs = env.reset()
    act = torch.zeros((E, U), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(n):
        if t and t % 100 == 0:
            obs = env.reset()
        if with_policy:
            act = policy(obs)
        obs = env.step(act)[0]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


run(20, True)
t_env = run(a.steps, False)
t_all = run(a.steps, True)
env.check()
print(f'{E} envs x {U} UE x {B} BS, actor 2x256 tanh ({a.dtype}), sampled actions')
print(f'env only          : {t_env * 1e3:8.3f} ms/step  {E / t_env:.3e} env-steps/s')
print(f'env + policy loop : {t_all * 1e3:8.3f} ms/step  {E / t_all:.3e} env-steps/s  ({E * U / t_all:.3e} agent-steps/s)')
print(f'share of the env in the loop: {100 * t_env / t_all:.1f} %')
