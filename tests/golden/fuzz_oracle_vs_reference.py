#!/usr/bin/env python3
"""Randomised pin of the CPU oracle against the REFERENCE ITSELF (build container only: needs /root/reference).

The committed fixtures pin the oracle on a fixed set of scenarios; this drives the reference's unmodified env classes
(behind the third-party stand-ins of _ref_shims.py, like gen_golden.py) and oracle/dcomp_oracle.c through random
configurations -- shapes, BS layouts, sharing models incl. max-cap, utilities, velocities, start positions incl. UEs parked
on a BS, rewards, agent kinds, fixed and continuing episodes -- with the same action tape, and applies the checks of
tests/test_oracle_golden.py::test_trajectory to every step: FP64 positions / waypoints / FSM / connection masks and
connection ORDER bit-exact, rates / utilities / observations / rewards to 1e-9.

    python tests/golden/fuzz_oracle_vs_reference.py [--cases 100] [--seed 0]
"""
import argparse
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, 'tools'))
sys.path.insert(0, os.path.join(REPO, 'tests'))

import gen_golden as G                                    # noqa: E402  (imports the reference behind the shims)
import fuzz_parity                                        # noqa: E402  (random_spec: plain-data configurations)
import test_oracle_golden as T                            # noqa: E402  (the oracle-side checks)
from oracle import oracle as orc                          # noqa: E402


def scenario_of(spec):
    scn = types.SimpleNamespace(width=spec['w'], height=spec['h'], bs_ids=[f'B{i}' for i in range(spec['B'])],
                                bs_pos=[tuple(p) for p in spec['bs_xy']], bs_sharing=list(spec['sh']))
    scn.ue_specs = [dict(id=str(i + 1), pos_x='random' if ix < 0 else ix, pos_y='random' if iy < 0 else iy, velocity=v,
                         util_func='log' if u == 0 else 'step', dr_req=rq)
                    for i, (v, u, rq, (ix, iy)) in enumerate(zip(spec['vel'], spec['util'], spec['req'], spec['init']))]
    if spec.get('pause'):                                 # RandomWaypoint(pause_duration, border_buffer), movement.py:87-104
        for s, pd, bb in zip(scn.ue_specs, spec['pause'], spec['border']):
            s['pause_duration'], s['border_buffer'] = pd, bb
    return scn


def run_one(spec):
    kind = spec['kind']
    steps = min(spec['steps'], 24)
    g = G.run_trajectory('fuzz', scenario_of(spec), kind, spec['seed'] % 100000, steps, reward=spec['reward'],
                         tape_mode='uniform' if spec['p_noop'] == 0.0 else 'sticky', rand_episodes=spec['rand_episodes'],
                         episodes=2, eps_len=steps, save=False)
    env, tape, U = T.make_env_from_fixture(g)
    k = int(g['cfg_kind'])
    t, consumed = 0, None
    for ep in range(2):
        env.set_tape(*tape.draw_episode(consumed))
        env.reset()
        T.check_snapshot(env, g, 'reset', ep, k)
        for _ in range(steps):
            env.step(g['actions'][t])
            T.check_snapshot(env, g, 'step', t, k)
            np.testing.assert_allclose(env.reward(), g['step_reward'][t], rtol=1e-9, atol=1e-12, err_msg=f'reward[{t}]')
            assert abs(env.sum_utility() - float(g['step_sum_utility'][t])) <= 1e-9 * max(1.0, abs(float(g['step_sum_utility'][t])))
            t += 1
        consumed = env.cursors()


def run_one_dynamic(spec):
    """UE arrival / departure: the checks of tests/test_oracle_golden.py::test_dynamic_ue_trajectory."""
    U = spec['U']
    spec = dict(spec, util=[0] * U, req=[1.0] * U, init=[[-1, -1]] * U)          # what dyn_setup() reconstructs from a record
    L = 45
    g = G.run_dynamic_trajectory('fuzz', scenario_of(spec), spec['kind'], spec['seed'] % 100000, L, ue_arrival=spec['arrival'],
                                 episodes=2, rand_episodes=spec['rand_episodes'], reward=spec['reward'], save=False)
    env, init_tape, new_tape, events, sched, U0, M = T.dyn_setup(g)
    kind = int(g['cfg_kind'])
    t, consumed, end_list = 0, None, None
    for ep in range(2):
        p0, t0 = init_tape.draw_episode(end_list, consumed)
        p1, t1 = new_tape.draw_episode()
        env.set_tape_ids(np.concatenate([p0, p1]), np.concatenate([t0, t1]))
        events.new_episode()
        env.reset()
        T.dyn_check(env, g, 'reset', ep, kind, M)
        for k in range(L):
            n_rem, n_add = sched[k]
            if n_rem or n_add:
                env.set_events(events.departures(n_rem, env.num_ue()), events.arrivals(n_add))
            env.step(g['actions'][t])
            T.dyn_check(env, g, 'step', t, kind, M)
            np.testing.assert_allclose(env.reward(), g['step_reward'][t], rtol=1e-9, atol=1e-12, err_msg=f'reward[{t}]')
            t += 1
        consumed, end_list = env.orig_consumed(), env.end_of_episode_list()


def has_maxcap_near_tie(spec):
    """Oracle-only dry run of run_one()'s trajectory: does a max-cap BS ever see its two closest connected UEs at squared
    distances that differ, but by less than 1e-9 (relative)?  There the FP64 rates of station.py:129-138 coincide and the
    winner is decided by connection order -- the case worth checking against the reference itself."""
    steps = 24
    init = [tuple(p) for p in spec['init']]
    env = orc.OracleEnv(spec['w'], spec['h'], spec['bs_xy'], spec['sh'], spec['vel'], kind=orc.MULTI if spec['kind'] == 'multi' else orc.CENTRAL,
                        ue_util=spec['util'], ue_dr_req=spec['req'], init_xy=init)
    tape = orc.RefRngTape(spec['seed'] % 100000, spec['w'], spec['h'], spec['vel'], init_xy=init, depth=64, rand_episodes=spec['rand_episodes'])
    acts = G.action_tape(2 * steps, spec['U'], spec['B'], 'sticky')
    t, consumed = 0, None
    for _ in range(2):
        env.set_tape(*tape.draw_episode(consumed))
        env.reset()
        for _ in range(steps):
            env.step(acts[t])
            t += 1
            st = env.state()
            for b in range(spec['B']):
                us = np.nonzero(st['conn'][:, b])[0]
                if spec['sh'][b] != 'max-cap' or len(us) < 2:
                    continue
                d2 = np.sort(((st['pos'][us] - np.array(spec['bs_xy'][b])) ** 2).sum(1))
                if d2[0] > 0 and 0 < (d2[1] - d2[0]) / d2[1] < 1e-9:
                    return True
        consumed = env.cursors()
    return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--near-ties', type=int, default=0, help='instead: find this many max-cap near-tie configurations with the oracle and check THEM against the reference')
    ap.add_argument('--cases', type=int, default=100)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--dump', default=None, help='--near-ties: write the configurations the reference confirmed to this JSON file (inputs only)')
    ap.add_argument('--max-pairs', type=int, default=240, help='U*B cap (the reference needs ~30 us per pair and step)')
    ap.add_argument('--many-stations', type=float, default=0.0, help='fraction of the cases with 33 ... 64 stations (fuzz_parity.many_stations; round 5)')
    ap.add_argument('--many-ues', type=float, default=0.0, help='fraction of the cases with 257 ... 1 024 UEs in one env (fuzz_parity.many_ues; round 5), at <= 6 stations and 10 steps: the reference is slow')
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    bad = done = 0
    if a.near_ties:
        tried, confirmed = 0, []
        while done < a.near_ties:
            spec = fuzz_parity.random_spec(rng)
            if 'max-cap' not in spec['sh'] or spec['U'] < 30 or spec['arrival']:
                continue
            spec.update(p_noop=0.5, steps=24)           # run_one(): sticky action tape, 2 x 24 steps
            tried += 1
            if not has_maxcap_near_tie(spec):
                continue
            done += 1
            try:
                run_one(spec)
                confirmed.append({k: (v.tolist() if hasattr(v, 'tolist') else v) for k, v in spec.items()})
            except AssertionError as ex:
                bad += 1
                print(f'near-tie case {done} FAILED: U={spec["U"]} B={spec["B"]} seed={spec["seed"]}\n   {str(ex)[:500]}', flush=True)
        if a.dump:
            import json
            with open(a.dump, 'w') as f:
                json.dump(confirmed, f, default=lambda o: o.tolist() if hasattr(o, 'tolist') else int(o))
        print(f'{done - bad} / {done} max-cap near-tie configurations (found among {tried} max-cap configurations): oracle == reference')
        sys.exit(1 if bad else 0)
    while done < a.cases:
        spec = fuzz_parity.many_ues(fuzz_parity.many_stations(fuzz_parity.random_spec(rng), a.many_stations), a.many_ues)
        if spec.get('many_ues'):                            # exempt from --max-pairs: few stations and steps instead
            nb = min(spec['B'], 6)
            spec.update(B=nb, bs_xy=spec['bs_xy'][:nb], sh=spec['sh'][:nb], steps=min(spec['steps'], 10))
        elif spec['U'] * spec['B'] > a.max_pairs:
            if not spec.get('many_stations'):
                continue
            n = max(1, a.max_pairs // spec['B'])           # keep the many-station cases: fewer UEs instead
            spec['U'] = n
            for k in ('vel', 'util', 'req', 'init', 'pause', 'border'):
                if spec.get(k) is not None:
                    spec[k] = spec[k][:n]
            spec['arrival'] = fuzz_parity._clamp_arrival(spec.get('arrival'), n, 250)     # (round 6: many-station cases keep their UE arrival / departure schedule)
        done += 1
        try:
            (run_one_dynamic if spec['arrival'] else run_one)(spec)
        except AssertionError as ex:
            bad += 1
            d = {k: spec[k] for k in ('kind', 'U', 'B', 'w', 'h', 'reward', 'seed', 'steps', 'rand_episodes', 'arrival')}
            print(f'case {done} FAILED: {d} sharing={sorted(set(spec["sh"]))}\n   {str(ex)[:500]}', flush=True)
    print(f'{a.cases - bad} / {a.cases} random configurations: oracle == reference')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
