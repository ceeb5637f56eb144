"""The connect / drop decision AT the range boundary (round 6), every step kernel against the oracle.

The reference decides `snr(sqrt(dx*dx + dy*dy)) > 2e-8` (station.py:122-127, 222-226).  tests/threshold_cases.py builds stations whose
reference-form squared distance to a UE is X - 2ulp ... X + 2ulp (X = the smallest double whose rounded root reaches d_T) and coordinates a few
doubles either side of that -- 1 000+ placements per kernel family -- for static UEs on integer points (the toggle of user.py:203-222 and the
drop of user.py:175-188 at the same position) and for MOVING UEs (a first pass records the trajectory, which does not depend on the
stations; the stations are then put on the threshold circle of positions the UEs hold after a move).  The oracle (literal reference form)
is the checker; connection masks must be bit-identical at every step.  The round-5 library (fused d^2 against fl(d_T^2)) fails these:
profiles/r06_threshold_red.txt.
"""
import zlib

import numpy as np
import pytest

from tests import parity
from tests import threshold_cases as tc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


# name, kind, U (listed UEs), B, E, environment switches, mode ('step' | 'rollout' | 'dyn'), what the dispatched kernel's name must contain
FAMILIES = [
    ('narrow_32x10', 'multi', 32, 10, 3, {}, 'step', 'step_kernel<10, 32'),
    ('narrow_16x32', 'central', 16, 32, 2, {}, 'step', 'step_kernel<32, 16'),
    # the kernels decide on a fused d^2 and redo a pair in the reference's form where the two can differ ((float)fused == (float)X);
    # DCOMP_DSQ_EXACT=1 is the path a host takes whose X sits too close to a float rounding boundary: EVERY pair in the reference's form
    ('narrow_32x10_always_exact', 'multi', 32, 10, 2, {'DCOMP_DSQ_EXACT': '1'}, 'step', 'step_kernel<10, 32'),
    ('tight_10x5', 'multi', 10, 5, 7, {'DCOMP_TIGHT': '1'}, 'step', 'step_kernel_tight<5'),
    ('tight_5x32', 'central', 5, 32, 5, {'DCOMP_TIGHT': '1'}, 'step', 'step_kernel_tight<32'),
    ('wide_128x32', 'multi', 128, 32, 2, {}, 'step', 'step_kernel_wide<32'),
    ('dyn_6x24', 'multi', 6, 24, 3, {}, 'dyn', 'step_kernel_dyn'),
    ('rollout_10x5', 'central', 10, 5, 6, {}, 'rollout', None),
    ('rollout_32x10', 'multi', 32, 10, 3, {}, 'rollout', None),
    ('big_32x40', 'multi', 32, 40, 2, {}, 'step', 'big_kernel'),
    ('big_16x64', 'central', 16, 64, 2, {}, 'step', 'big_kernel'),
    ('big_forced_32x10', 'multi', 32, 10, 2, {'DCOMP_FORCE_BIG': '1'}, 'step', 'big_kernel'),
    # the generic kernel's other instantiations compile the same decision again: the fused rollout (ROLL) and the event phase (DYN)
    ('big_rollout_12x40', 'multi', 12, 40, 2, {}, 'rollout', 'big_kernel'),
    ('big_rollout_forced_10x5', 'central', 10, 5, 3, {'DCOMP_FORCE_BIG': '1'}, 'rollout', 'big_kernel'),
    ('big_dyn_6x40', 'multi', 6, 40, 3, {}, 'dyn', 'big_kernel'),
]
ARRIVAL = {1: 2, 3: -1, 5: 1, 8: -2}            # UE arrival / departure while the threshold decisions are taken (base.py:433-443)


def _slots(U, mode):
    """UE slots per env: the list, or max_ues = the largest simultaneous count of the schedule (base.py:79-84)."""
    if mode != 'dyn':
        return U
    n = peak = 0
    for t in sorted(ARRIVAL):
        n += ARRIVAL[t]
        peak = max(peak, n)
    return U + peak


def _d_t():
    from oracle import oracle as orc
    return orc.connect_threshold_distance()


def _scenario(B, W, H, bs_pos, specs, sharing='mixed'):
    from deepcomp_amd import scenarios
    scn = scenarios.grid_map(B, sharing)
    scn.width, scn.height = W, H
    scn.bs_pos = [(float(x), float(y)) for x, y in bs_pos]
    scn.ue_specs = specs
    return scn


def _specs(vels, xy=None):
    return [dict(id=str(i + 1), pos_x='random' if xy is None else int(xy[i][0]), pos_y='random' if xy is None else int(xy[i][1]),
                 velocity=v, util_func='log', dr_req=1) for i, v in enumerate(vels)]


def _oracle(scn, kind, E, seed, max_ues=None):
    from oracle import oracle as orc
    init_xy = [(s['pos_x'], s['pos_y']) if s['pos_x'] != 'random' else (-1, -1) for s in scn.ue_specs]
    envs = []
    for e in range(E):
        o = orc.OracleEnv(int(scn.width), int(scn.height), scn.bs_pos, scn.bs_sharing, [s['velocity'] for s in scn.ue_specs],
                          kind=orc.MULTI if kind == 'multi' else orc.CENTRAL, init_xy=init_xy, max_ues=max_ues)
        o.set_philox(seed, e)
        envs.append(o)
    return orc.OracleBatch(envs)


def _core(torch, scn, kind, E, seed, mode):
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    m, bs, ues = build_from_scenario(scn)
    return BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=seed, rng='philox', rand_episodes=True,
                            ue_arrival=dict(ARRIVAL) if mode == 'dyn' else None)


def _run(torch, scn, kind, E, seed, acts, mode, kernel_tag, msg):
    """reset + len(acts) steps of the HIP path against the oracle: masks / positions bit-exact at every step (plus every float bar of
    tests/parity.py).  acts: [T, E, slots] uint8."""
    from oracle import oracle as orc
    core = _core(torch, scn, kind, E, seed, mode)
    M, B, T = core.U, core.B, acts.shape[0]
    assert M == acts.shape[2], f'{msg}: {M} slots, actions for {acts.shape[2]}'
    if kernel_tag is not None:
        assert kernel_tag in core.step_kernel_name, f'{msg}: dispatched {core.step_kernel_name}, expected {kernel_tag}'
    ob = _oracle(scn, kind, E, seed, max_ues=M if mode == 'dyn' else None)
    sched = orc.arrival_schedule(100, ARRIVAL) if mode == 'dyn' else None
    core.reset()
    o_reset = ob.reset()
    if mode == 'rollout':
        assert core.rollout_is_fused(T), f'{msg}: this batch does not take the fused rollout kernel'
        obs_shape = (T,) + tuple(core.obs.shape)
        out = {'obs': torch.empty(obs_shape, device='cuda'), 'reward': torch.empty((T,) + tuple(core.reward.shape), device='cuda'),
               'ue_dr': torch.empty((T, E, M), device='cuda'), 'ue_utility': torch.empty((T, E, M), device='cuda')}
        core.rollout(torch.from_numpy(acts).cuda(), out=out)
        core.check()
        got_obs = out['obs'].cpu().numpy()
        got_dr, got_ut = out['ue_dr'].cpu().numpy(), out['ue_utility'].cpu().numpy()
        for t in range(T):
            o_obs, o_rew, o_conn, o_pos = ob.step(acts[t])
            r = parity.assert_rates(core, ob, f'{msg} step {t}', ue_dr=got_dr[t], ue_utility=got_ut[t], ewma=False)
            parity.assert_obs(got_obs[t], o_obs, kind, M, B, dr_rel=r['dr_rel'], msg=f'{msg} step {t}')      # `connected` exact
        st = core.state_host()
        assert np.array_equal(st['pos'], o_pos) and np.array_equal(st['conn'], o_conn), f'{msg}: final state'
        return
    parity.assert_obs(core.obs.cpu().numpy(), o_reset, kind, M, B, msg=f'{msg} reset')
    for t in range(T):
        if sched is not None:
            n_rem, n_add = sched[t]
            if n_rem or n_add:
                for o in ob.envs:
                    o.set_event_counts(n_rem, n_add)
        core.step(torch.from_numpy(acts[t]).cuda())
        o_obs, o_rew, o_conn, o_pos = ob.step(acts[t])
        if mode == 'dyn':
            st = core.state_host()
            assert np.array_equal(st['conn'], o_conn), f'{msg} step {t}: connection masks differ'
            assert np.array_equal(st['pos'], o_pos), f'{msg} step {t}: positions'
            parity.assert_obs(core.obs.cpu().numpy(), o_obs, kind, M, B, msg=f'{msg} step {t}')
        else:
            parity.assert_step(core, ob, o_obs, o_rew, o_conn, o_pos, kind, msg=f'{msg} step {t}')
    core.check()


@pytest.mark.parametrize('fam', FAMILIES, ids=[f[0] for f in FAMILIES])
def test_static_ues_at_the_threshold_circle(torch_cuda, fam, monkeypatch):
    """>= 1 000 placements per kernel family: static UEs on integer points, one station per placement within doubles of the boundary.
    Script: every UE asks for each of its stations (connect decided at the boundary), a no-op step (drop decided at the boundary), then two
    more rounds of toggles (disconnect + re-connect).  Masks bit-identical to the oracle's at every step."""
    name, kind, U, B, E, envvars, mode, tag = fam
    for k, v in envvars.items():
        monkeypatch.setenv(k, v)
    d_t = _d_t()
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    n_cfg = -(-1024 // B)
    S = -(-B // U)
    n_place = n_diff = n_edge = 0
    for c in range(n_cfg):
        W, H = int(rng.integers(120, 700)), int(rng.integers(120, 700))
        ue_xy, bs_pos, tgt, info = tc.static_case(rng, U, B, W, H, d_t)
        n_place += B
        n_diff += info['differs_from_round5_predicate']
        n_edge += info['within_one_ulp_of_X']
        # (max-cap stations take neither the tight nor the wide kernel)
        sharing = (('mixed', 'rate-fair', 'resource-fair', 'proportional-fair') if ('wide' in name or 'tight' in name)
                   else ('mixed', 'resource-fair', 'max-cap', 'proportional-fair'))[c % 4]
        scn = _scenario(B, W, H, bs_pos, _specs([0] * U, ue_xy), sharing)
        M = _slots(U, mode)
        rounds = []
        for s in range(S):
            a = np.zeros((E, M), dtype=np.uint8)
            for u in range(U):
                b = u + s * U
                if b < B:
                    a[:, u] = b + 1
            rounds.append(a)
        acts = np.stack(rounds + [np.zeros((E, M), dtype=np.uint8)] + rounds + rounds)
        _run(torch_cuda, scn, kind, E, 100 + c, acts, mode, tag, f'{name} cfg {c}')
    assert n_place >= 1000
    # the placements must be able to tell the two predicates apart: many sit where fma(dy, dy, dx*dx) < fl(d_T^2) decides differently
    assert n_diff >= n_place // 20 and n_edge >= n_place // 5, (n_place, n_diff, n_edge)


@pytest.mark.parametrize('fam', FAMILIES, ids=[f[0] for f in FAMILIES])
def test_moving_ues_cross_the_threshold_circle(torch_cuda, fam, monkeypatch):
    """Stations on the threshold circle of positions UEs hold AFTER a move (non-integer coordinates): the drop decision of that step and the
    connect decision of the next one are taken within doubles of the boundary.  Pass 1 (oracle alone) records the trajectory -- movement does
    not depend on the stations --, pass 2 runs HIP and oracle with the stations in place."""
    name, kind, U, B, E, envvars, mode, tag = fam
    for k, v in envvars.items():
        monkeypatch.setenv(k, v)
    d_t = _d_t()
    X = tc.boundary_q(d_t)
    rng = np.random.default_rng(zlib.crc32((name + 'mv').encode()))
    T = 12
    n_cfg = max(4, -(-256 // B))
    n_dec = n_diff = 0
    from oracle import oracle as orc
    for c in range(n_cfg):
        W, H = int(rng.integers(150, 500)), int(rng.integers(150, 500))
        vels = [(1, 2, 3, 'slow', 'fast', 2.5)[int(rng.integers(6))] for _ in range(U)]
        dummy = [(float(rng.uniform(0, W)), float(rng.uniform(0, H))) for _ in range(B)]
        scn0 = _scenario(B, W, H, dummy, _specs(vels))
        seed = 500 + c
        M = _slots(U, mode)
        ob = _oracle(scn0, kind, E, seed, max_ues=M if mode == 'dyn' else None)
        sched = orc.arrival_schedule(100, ARRIVAL) if mode == 'dyn' else None
        ob.reset()
        traj = [np.stack([o.state()['pos'] for o in ob.envs])]
        zero = np.zeros((E, M), dtype=np.uint8)
        for t in range(T):
            if sched is not None and (sched[t][0] or sched[t][1]):
                for o in ob.envs:
                    o.set_event_counts(*sched[t])
            ob.step(zero, want_obs=False)
            traj.append(np.stack([o.state()['pos'] for o in ob.envs]))
        traj = np.stack(traj)[:, :, :U]                 # (UE arrival / departure: the first U slots hold a UE at every step of ARRIVAL)
        bs_pos, act_small, nd = tc.moving_case(traj, rng, B, d_t)
        acts = np.zeros((T, E, M), dtype=np.uint8)
        acts[:, :, :act_small.shape[2]] = act_small
        n_dec += nd
        for b in range(B):
            # how many of the boundary positions the round-5 predicate would have decided differently
            d = np.abs([tc.q_ref(*p, *bs_pos[b]) - X for p in traj.reshape(-1, 2)])
            p = traj.reshape(-1, 2)[int(np.argmin(d))]
            ref, old, _ = tc.classify(p[0], p[1], bs_pos[b][0], bs_pos[b][1], d_t, X)
            n_diff += int(ref != old)
        scn = _scenario(B, W, H, bs_pos, _specs(vels))
        _run(torch_cuda, scn, kind, E, seed, acts, mode, tag, f'{name} moving cfg {c}')
    assert n_dec >= 200 and n_diff >= 10, (n_dec, n_diff)
