#!/usr/bin/env python3
"""Rate of the RLlib PROTOCOL path (deepcomp_amd/rllib_adapter.py) -- what an unmodified RLlib 1.4 sampler calls
(simulation.py:143 -> RolloutWorker -> VectorEnv.vector_step / BaseEnv.poll + send_actions) -- next to the tensor path.

    python tools/adapter_rate.py [--seconds 2.0] [--out gpurun_out/r05_adapter_rate.txt]

Per (env kind, E): env-steps/s of
  before      per-step dict building over a fresh host array (the round-4 code path: env_config['persistent_views'] = False)
  persistent  the dicts of views built ONCE over the env's pinned host buffer (round 5 default)
  + flatten   the same, with the minimal consumer RLlib has: every observation dict flattened in sorted-key order (flatten_obs)
  info=none   central only: without the per-UE metric dicts of info (env_config['info_level'])
  flat_obs    env_config['flat_obs'] = True (round 6): observation_space is the flattened Box, the protocol methods hand out rows of one pinned
              array; the consumer is what RLlib's preprocessor for a Box does -- NoPreprocessor.transform returns its argument -- called once per
              observation (a Python call per env / per agent: at 1 024 x 32 agents that call alone is ~1.5 ms per step)
  tensors     poll_tensors() / send_action_tensor(): device tensors, no per-env objects (upper bound; a torch.randint policy on the device)
Actions are random, generated outside the timed loop.  Not a product path; numbers go to INTEGRATION.md section 1.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from deepcomp_amd import scenarios  # noqa: E402
from deepcomp_amd.entities import build_from_scenario  # noqa: E402
from deepcomp_amd.rllib_adapter import CentralVectorEnv, MultiAgentBaseEnv, flatten_obs  # noqa: E402


def cfg(U, B, E, **kw):
    m, bs, ues = build_from_scenario(scenarios.grid_map(B, 'mixed').with_ues(num_slow=U))
    c = dict(map=m, bs_list=bs, ue_list=ues, seed=42, episode_length=100, reward='avg', rand_episodes=True, num_envs=E, rng='philox')
    c.update(kw)
    return c


def no_preprocessor(o):
    """RLlib's NoPreprocessor.transform (what it picks for a Box space): the observation itself."""
    return o


def loop(step, seconds):
    step(0)
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        step(n)
        n += 1
    return n / (time.perf_counter() - t0)


def central(U, B, E, seconds, flatten=False, tensors=False, **kw):
    env = CentralVectorEnv(cfg(U, B, E, **kw))
    env.vector_reset()
    rng = np.random.default_rng(1)
    pool = [rng.integers(0, B + 1, size=(E, U)).tolist() for _ in range(8)]
    dev_pool = torch.randint(0, B + 1, (8, E, U), device='cuda', dtype=torch.uint8)

    def step(n):
        if n and n % 100 == 0:
            env.vector_reset()
        if tensors:
            env.send_action_tensor(dev_pool[n & 7])
            if n % 50 == 49:
                torch.cuda.synchronize()
            return
        obs, rew, dones, infos = env.vector_step(pool[n & 7])
        if flatten:
            for o in obs:
                flatten_obs(o)
        elif kw.get('flat_obs') and not os.environ.get('ADAPTER_RATE_NO_TOUCH'):
            for o in obs:
                no_preprocessor(o)
    r = loop(step, seconds)
    torch.cuda.synchronize()
    return r * E


def multi(U, B, E, seconds, flatten=False, tensors=False, **kw):
    env = MultiAgentBaseEnv(cfg(U, B, E, **kw))
    ids = env.agent_ids
    rng = np.random.default_rng(1)
    pool = [{e: dict(zip(ids, row)) for e, row in enumerate(rng.integers(0, B + 1, size=(E, U)).tolist())} for _ in range(8)]
    dev_pool = torch.randint(0, B + 1, (8, E, U), device='cuda', dtype=torch.uint8)
    env.poll()

    def step(n):
        if n and n % 100 == 0:
            env.try_reset(0)
            env.poll()
        if tensors:
            env.send_action_tensor(dev_pool[n & 7])
            if n % 50 == 49:
                torch.cuda.synchronize()
            return
        env.send_actions(pool[n & 7])
        obs, rew, dones, infos, _ = env.poll()
        if flatten:
            for per_env in obs.values():
                for o in per_env.values():
                    flatten_obs(o)
        elif kw.get('flat_obs') and not os.environ.get('ADAPTER_RATE_NO_TOUCH'):
            for per_env in obs.values():
                for o in per_env.values():
                    no_preprocessor(o)
    r = loop(step, seconds)
    torch.cuda.synchronize()
    return r * E


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=2.0)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    lines = ['# tools/adapter_rate.py: env-steps/s through the RLlib protocol adapters (one MI355X, one host thread)',
             f'# {torch.cuda.get_device_name(0)}, torch {torch.__version__}, {a.seconds:.1f} s per cell',
             '| env | E | before (fresh dicts per step) | persistent views | ... + flatten_obs of every dict | info_level=none | flat_obs rows + the no-op preprocessor per observation | flat_obs, rows not touched | tensor path (no per-env objects) |',
             '|---|---|---|---|---|---|---|---|---|']
    for name, fn, U, B in (('central 10 x 5 (VectorEnv.vector_step)', central, 10, 5), ('multi 32 x 10 (BaseEnv.poll + send_actions)', multi, 32, 10)):
        for E in (1, 16, 256, 1024):
            before = fn(U, B, E, a.seconds, persistent_views=False)
            pers = fn(U, B, E, a.seconds)
            flat = fn(U, B, E, a.seconds, flatten=True)
            none = fn(U, B, E, a.seconds, info_level='none') if fn is central else None
            tens = fn(U, B, E, a.seconds, tensors=True)
            kw = {'info_level': 'none'} if fn is central else {}
            rows = fn(U, B, E, a.seconds, flat_obs=True, **kw)
            os.environ['ADAPTER_RATE_NO_TOUCH'] = '1'
            rows0 = fn(U, B, E, a.seconds, flat_obs=True, **kw)
            del os.environ['ADAPTER_RATE_NO_TOUCH']
            lines.append(f'| {name} | {E} | {before:,.0f} | {pers:,.0f} ({pers / before:.2f} x) | {flat:,.0f} | {"-" if none is None else f"{none:,.0f}"} | {rows:,.0f} | {rows0:,.0f} | {tens:,.0f} |')
            print(lines[-1], flush=True)
    txt = '\n'.join(lines) + '\n'
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, 'w').write(txt)
    print(txt)


if __name__ == '__main__':
    main()
