"""The CPU oracle (oracle/dcomp_oracle.c) against the golden fixtures recorded from the reference
itself (tests/golden/gen_golden.py).  Indices / masks / FSM state / FP64 positions: exact.
Other FP64 floats: 1e-9 relative or tighter (both sides are FP64; libm vs numpy differ in the last
ulps, and log2(1+snr) itself only carries ~8 significant digits near the connect threshold)."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
RTOL = 1e-12


def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


# ------------------------------------------------------------------ G1-G3 known-answer tables
def test_channel_table():
    g = load('channel')
    for d, s, c, r in zip(g['d'], g['snr'], g['can_connect'], g['dr_unshared']):
        assert orc.snr(d) == pytest.approx(s, rel=RTOL)
        assert orc.can_connect(d) == bool(c)
        assert orc.dr_unshared(d) == pytest.approx(r, rel=1e-9)
    for (x, y), s, c in zip(g['xy'], g['snr_xy'], g['can_xy']):
        d = np.sqrt(x * x + y * y)
        assert orc.snr(d) == pytest.approx(s, rel=RTOL)
        assert orc.can_connect(d) == bool(c)


def test_survey_appendix_b_values():
    # SURVEY.md Appendix B (probe values of the reference arithmetic)
    assert orc.snr(1.0) == pytest.approx(3.232562595879e-02, rel=1e-12)
    assert orc.snr(0.0) == pytest.approx(3.502202856706e+52, rel=1e-12)
    assert orc.dr_unshared(46.0) == pytest.approx(1.017513972, rel=1e-9)
    assert orc.can_connect(68.0) and not orc.can_connect(69.0)
    assert orc.connect_threshold_distance() == pytest.approx(68.92488308058013, abs=1e-9)


def test_utility_table():
    g = load('utility')
    for dr, lu, su in zip(g['dr'], g['log_utility'], g['step_utility_req1']):
        assert orc.log_utility(dr) == pytest.approx(lu, rel=RTOL, abs=1e-13)
        assert orc.step_utility(dr, 1) == su


def _connect_in_order(env, n_connected):
    n = env.U
    for u in range(n_connected):
        a = np.zeros(n, np.int32)
        a[u] = 1                      # action 1 = toggle BS 0 (base.py:259-263)
        env.step(a)


def test_sharing_table():
    """Every row of the reference's sharing-model table (all four models incl. proportional-fair, with and without EWMA
    history): 1 BS at the origin, static UEs on the x axis, connected oldest-first.  `dr_last_unconnected`: all but the last UE
    connected and everyone asked -- the last one takes the temporarily-appended path of station.py:164-168;
    `dr_all_connected`: everyone connected."""
    g = load('sharing')
    checked = 0
    for i in range(len(g['model'])):
        n, model = int(g['n'][i]), int(g['model'][i])
        dist, ewma = g['dist'][i][:n], g['ewma'][i][:n]
        for column, k_conn in (('dr_last_unconnected', n - 1), ('dr_all_connected', n)):
            env = orc.OracleEnv(400, 400, [(0, 0)], [model], [0] * n, kind=orc.MULTI, init_xy=[(int(d), 0) for d in dist])
            env.set_tape(np.zeros((n, 2), np.int32), np.tile(np.array([0, 200, 200], np.int32), (n, 4, 1)))
            env.reset()
            _connect_in_order(env, k_conn)
            got = [env.probe_data_rate(0, u, ewma) for u in range(n)]              # the stepping above moved the EWMAs: reseed them
            np.testing.assert_allclose(got, g[column][i][:n], rtol=1e-9, atol=0, err_msg=f'row {i} {column} model {model}')
            checked += 1
    assert checked == 2 * len(g['model']) >= 60


# ------------------------------------------------------------------ G4 movement traces
@pytest.mark.parametrize('name', ['slow', 'fast', 'static', 'fixed4', 'slow_small'])
def test_movement_trace(name):
    g = load('movement')
    tr = g[name + '_trace']
    w, h, vel, seed = (int(x) for x in g[name + '_cfg'])
    # User.seed(seed) seeds both streams with `seed` itself (user.py:94-96): base seed = seed - 100
    tape = orc.RefRngTape(seed - 100, w, h, [vel], depth=120)
    pos0, trip = tape.draw_episode()
    env = orc.OracleEnv(w, h, [(0, 0)], ['resource-fair'], [vel])
    env.set_tape(pos0, trip)
    env.reset()
    for t in range(tr.shape[0]):
        s = env.state()
        got = [s['pos'][0, 0], s['pos'][0, 1], s['wp'][0, 0], s['wp'][0, 1], s['vel'][0], s['pausing'][0],
               s['curr_pause'][0]]
        assert got == list(tr[t]), f"step {t}: {got} != {list(tr[t])}"     # bit-exact FP64 positions
        env.step(np.zeros(1, np.int32))
    assert env.cursors()[0] <= 120


# ------------------------------------------------------------------ G5/G6 trajectories
TRAJ = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'traj_*.npz'))) + \
       sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'estack_*.npz')))


def make_env_from_fixture(g, depth=64):
    U = g['cfg_ue_vel'].shape[0]
    w, h = (int(x) for x in g['cfg_map_wh'])
    vel = [int(v) for v in g['cfg_ue_vel']]
    if 'cfg_ue_vel_num' in (g.files if hasattr(g, 'files') else g):                  # fixed velocities that are no integers (movement.py:116-117)
        vel = [float(n) if n >= 0 else v for v, n in zip(vel, g['cfg_ue_vel_num'])]
    init = [tuple(int(v) for v in xy) for xy in g['cfg_ue_init_xy']]
    pause = [int(v) for v in g['cfg_ue_pause']] if 'cfg_ue_pause' in (g.files if hasattr(g, 'files') else g) else None      # RandomWaypoint parameters away
    border = [int(v) for v in g['cfg_ue_border']] if 'cfg_ue_border' in (g.files if hasattr(g, 'files') else g) else None   # from the defaults (movement.py:87)
    env = orc.OracleEnv(w, h, g['cfg_bs_pos'], list(g['cfg_bs_sharing']), vel, kind=int(g['cfg_kind']),
                        reward_agg=int(g['cfg_reward']), ue_util=g['cfg_ue_util'], ue_dr_req=g['cfg_ue_dr_req'], init_xy=init,
                        pause=pause, border=border)
    tape = orc.RefRngTape(int(g['cfg_seed']), w, h, vel, init_xy=init, depth=depth,
                          rand_episodes=bool(g['cfg_rand_episodes']), border=border)
    return env, tape, U


def check_snapshot(env, g, prefix, i, kind):
    s = env.state()
    o = env.obs()
    for k in ('pos', 'wp', 'vel'):
        assert np.array_equal(s[k], g[f'{prefix}_{k}'][i]), f'{prefix}[{i}] {k} not bit-exact'
    for k in ('pausing', 'curr_pause', 'conn'):
        assert np.array_equal(s[k], g[f'{prefix}_{k}'][i]), f'{prefix}[{i}] {k}'
    assert np.array_equal(s['conn_order'], g[f'{prefix}_conn_order'][i]), f'{prefix}[{i}] conn_order'
    for k in ('dr', 'curr_dr', 'ewma', 'utility'):
        np.testing.assert_allclose(s[k], g[f'{prefix}_{k}'][i], rtol=1e-9, atol=1e-12, err_msg=f'{prefix}[{i}] {k}')
    assert np.array_equal(o['connected'], g[f'{prefix}_obs_connected'][i])
    np.testing.assert_allclose(o['dr'], g[f'{prefix}_obs_dr'][i], rtol=RTOL, atol=1e-300)
    np.testing.assert_allclose(o['utility'], g[f'{prefix}_obs_utility'][i], rtol=1e-9, atol=1e-12)
    if kind == orc.MULTI:
        np.testing.assert_allclose(o['ues_at_bs'], g[f'{prefix}_obs_ues_at_bs'][i], rtol=RTOL)
        np.testing.assert_allclose(o['util_at_bs'], g[f'{prefix}_obs_util_at_bs'][i], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('name', TRAJ)
def test_trajectory(name):
    g = load(name)
    env, tape, U = make_env_from_fixture(g)
    kind = int(g['cfg_kind'])
    episodes = int(g['cfg_episodes'])
    steps_per_ep = g['actions'].shape[0] // episodes
    t = 0
    consumed = None
    for ep in range(episodes):
        pos0, trip = tape.draw_episode(consumed)
        env.set_tape(pos0, trip)
        env.reset()
        check_snapshot(env, g, 'reset', ep, kind)
        for _ in range(steps_per_ep):
            env.step(g['actions'][t])
            check_snapshot(env, g, 'step', t, kind)
            np.testing.assert_allclose(env.reward(), g['step_reward'][t], rtol=1e-9, atol=1e-12,
                                       err_msg=f'reward[{t}]')
            assert env.sum_utility() == pytest.approx(float(g['step_sum_utility'][t]), rel=1e-9, abs=1e-12)
            assert env.time() == int(g['step_time'][t])
            t += 1
        consumed = env.cursors()
        assert consumed.max() <= trip.shape[1]


RESEED = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'reseed_*.npz')))


@pytest.mark.parametrize('name', RESEED)
def test_seed_on_a_live_env(name):
    """MobileEnv.seed() while an episode runs and right before a reset (base.py:132-143, 171-173), reference-run fixtures:
    the running episode continues on the start of the new streams; reset() of a rand_episodes=False env re-seeds with the
    CONFIGURED seed, a rand_episodes=True env keeps the new streams."""
    g = load(name)
    assert RESEED and 'cfg_seed_at' in g.files
    env, tape, U = make_env_from_fixture(g)
    kind = int(g['cfg_kind'])
    episodes = int(g['cfg_episodes'])
    steps_per_ep = g['actions'].shape[0] // episodes
    seed_at = {int(t): int(s) for t, s in g['cfg_seed_at']}
    before_reset = {int(e): int(s) for e, s in g['cfg_seed_before_reset']} if 'cfg_seed_before_reset' in g.files else {}
    t, consumed, pos0 = 0, None, None
    for ep in range(episodes):
        if ep in before_reset:
            tape.reseed_live(before_reset[ep], consumed)
        pos0, trip = tape.draw_episode(consumed)
        env.set_tape(pos0, trip)
        env.reset()
        check_snapshot(env, g, 'reset', ep, kind)
        for _ in range(steps_per_ep):
            if t in seed_at:
                env.set_tape(pos0, tape.reseed_live(seed_at[t], env.cursors()))
            env.step(g['actions'][t])
            check_snapshot(env, g, 'step', t, kind)
            np.testing.assert_allclose(env.reward(), g['step_reward'][t], rtol=1e-9, atol=1e-12, err_msg=f'reward[{t}]')
            t += 1
        consumed = env.cursors()


def test_bad_action_rejected():
    env = orc.OracleEnv(150, 100, [(50, 50), (100, 50)], ['resource-fair'] * 2, ['slow'] * 2)
    env.set_philox(1, 0)
    env.reset()
    with pytest.raises(AssertionError):
        env.step([3, 0])


def test_philox_known_answers():
    # Random123 known-answer vectors for Philox4x32-10
    assert orc.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert orc.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


# ------------------------------------------------------------------ G8 UE arrival / departure (base.py:433-443, 592-618)
DYN = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'dyn_*.npz')))


def dyn_setup(g, depth=48):
    """Oracle env + the reference's draws for a dynamic-UE fixture.  Tapes: initial UEs by list position, then one
    'slow' tape per possible id of an arriving UE (seed + 100*id, base.py:602-604)."""
    w, h = (int(x) for x in g['cfg_map_wh'])
    vel = [int(v) for v in g['cfg_ue_vel']]
    if 'cfg_ue_vel_num' in (g.files if hasattr(g, 'files') else g):                  # fixed velocities that are no integers (movement.py:116-117)
        vel = [float(n) if n >= 0 else v for v, n in zip(vel, g['cfg_ue_vel_num'])]
    U0, M, seed = len(vel), int(g['cfg_max_ues']), int(g['cfg_seed'])
    L = int(g['cfg_eps_len'])
    arr = {int(t): int(n) for t, n in zip(g['cfg_arrival_t'], g['cfg_arrival_n'])} or None
    interval = int(g['cfg_new_ue_interval'])
    sched = orc.arrival_schedule(L, arr, interval if interval > 0 else None)
    max_id = U0 + sum(a for _, a in sched)
    env = orc.OracleEnv(w, h, g['cfg_bs_pos'], list(g['cfg_bs_sharing']), vel, kind=int(g['cfg_kind']),
                        reward_agg=int(g['cfg_reward']), max_ues=M)
    init_tape = orc.DynRefStreams(seed, w, h, vel, depth=depth, rand_episodes=bool(g['cfg_rand_episodes']))
    new_tape = orc.RefRngTape(seed, w, h, ['slow'] * max_id, depth=depth)      # re-seeded at every arrival
    events = orc.RefEventDraws(seed, w, h, rand_episodes=bool(g['cfg_rand_episodes']))
    return env, init_tape, new_tape, events, sched, U0, M


def dyn_check(env, g, prefix, i, kind, M):
    n = int(g[f'{prefix}_num_ue'][i])
    assert env.num_ue() == n
    assert np.array_equal(env.uids(), g[f'{prefix}_ue_ids'][i])
    s, o = env.state(), env.obs()
    for k in ('pos', 'wp', 'vel', 'pausing', 'curr_pause', 'conn', 'conn_order'):
        assert np.array_equal(s[k], g[f'{prefix}_{k}'][i]), f'{prefix}[{i}] {k}'
    for k in ('dr', 'curr_dr', 'ewma', 'utility'):
        np.testing.assert_allclose(s[k], g[f'{prefix}_{k}'][i], rtol=1e-9, atol=1e-12, err_msg=f'{prefix}[{i}] {k}')
    assert np.array_equal(o['connected'], g[f'{prefix}_obs_connected'][i])
    np.testing.assert_allclose(o['dr'], g[f'{prefix}_obs_dr'][i], rtol=RTOL, atol=1e-300)
    np.testing.assert_allclose(o['utility'], g[f'{prefix}_obs_utility'][i], rtol=1e-9, atol=1e-12)
    if kind == orc.MULTI:
        np.testing.assert_allclose(o['ues_at_bs'], g[f'{prefix}_obs_ues_at_bs'][i], rtol=RTOL)
        np.testing.assert_allclose(o['util_at_bs'], g[f'{prefix}_obs_util_at_bs'][i], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('name', DYN)
def test_dynamic_ue_trajectory(name):
    g = load(name)
    env, init_tape, new_tape, events, sched, U0, M = dyn_setup(g)
    kind, episodes, L = int(g['cfg_kind']), int(g['cfg_episodes']), int(g['cfg_eps_len'])
    t, consumed, end_list = 0, None, None
    for ep in range(episodes):
        p0, t0 = init_tape.draw_episode(end_list, consumed)
        p1, t1 = new_tape.draw_episode()
        env.set_tape_ids(np.concatenate([p0, p1]), np.concatenate([t0, t1]))
        events.new_episode()
        env.reset()
        dyn_check(env, g, 'reset', ep, kind, M)
        for k in range(L):
            n_rem, n_add = sched[k]
            if n_rem or n_add:
                env.set_events(events.departures(n_rem, env.num_ue()), events.arrivals(n_add))
            env.step(g['actions'][t])
            dyn_check(env, g, 'step', t, kind, M)
            np.testing.assert_allclose(env.reward(), g['step_reward'][t], rtol=1e-9, atol=1e-12, err_msg=f'reward[{t}]')
            assert env.sum_utility() == pytest.approx(float(g['step_sum_utility'][t]), rel=1e-9, abs=1e-12)
            t += 1
        consumed, end_list = env.orig_consumed(), env.end_of_episode_list()


RESEED_DYN = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'reseeddyn_*.npz')))


@pytest.mark.parametrize('name', RESEED_DYN)
def test_seed_on_a_live_env_with_ue_arrival(name):
    """MobileEnv.seed() mid-episode while UEs arrive / depart (base.py:132-143): listed UEs -- initial and arrived -- are re-seeded
    by list position, departures / arrival points restart from the new seed, later arrivals keep the configured seed."""
    g = load(name)
    assert RESEED_DYN
    env, init_tape, new_tape, events, sched, U0, M = dyn_setup(g)
    kind, episodes, L = int(g['cfg_kind']), int(g['cfg_episodes']), int(g['cfg_eps_len'])
    seed_at = {int(t): int(s) for t, s in g['cfg_seed_at']}
    t, consumed, end_list = 0, None, None
    for ep in range(episodes):
        p0, t0 = init_tape.draw_episode(end_list, consumed)
        p1, t1 = new_tape.draw_episode()
        pos_all, trip_all = np.concatenate([p0, p1]), np.concatenate([t0, t1])
        env.set_tape_ids(pos_all, trip_all)
        events.new_episode()
        env.reset()
        dyn_check(env, g, 'reset', ep, kind, M)
        for k in range(L):
            if t in seed_at:
                cur = env.cursors()
                slots = [(uid, born, int(cur[s_])) for s_, (uid, born) in enumerate(env.end_of_episode_list())]
                trip_all = init_tape.reseed_live(seed_at[t], trip_all, slots)
                events.reseed_live(seed_at[t])
                env.set_tape_ids(pos_all, trip_all)
            n_rem, n_add = sched[k]
            if n_rem or n_add:
                env.set_events(events.departures(n_rem, env.num_ue()), events.arrivals(n_add))
            env.step(g['actions'][t])
            dyn_check(env, g, 'step', t, kind, M)
            np.testing.assert_allclose(env.reward(), g['step_reward'][t], rtol=1e-9, atol=1e-12, err_msg=f'reward[{t}]')
            t += 1
        consumed, end_list = env.orig_consumed(), env.end_of_episode_list()


# ------------------------------------------------------------------ G9 single-agent env (base.py:227-245, 350-369)
@pytest.mark.parametrize('name,scn_args', [('single_custom3x4_s42', ('custom', 2, 1)), ('single_small2x2_s43', ('small', 2, 0))])
def test_single_agent_env(name, scn_args):
    """'--agent single' -> RelNormEnv: one UE acts per step (round robin), obs of the next UE, reward of the acting UE."""
    from deepcomp_amd import scenarios
    g = load(name)
    scn = scenarios.get_scenario(scn_args[0], 'mixed').with_ues(num_slow=scn_args[1], num_fast=scn_args[2])
    vel = [s['velocity'] for s in scn.ue_specs]
    U, B = len(vel), scn.num_bs
    env = orc.OracleEnv(int(scn.width), int(scn.height), scn.bs_pos, scn.bs_sharing, vel, kind=orc.MULTI)
    tape = orc.RefRngTape(int(g['cfg_seed']), int(scn.width), int(scn.height), vel, depth=40)
    env.set_tape(*tape.draw_episode())
    env.reset()

    def check_obs(t):
        o, i = env.obs(), env.time() % U
        assert np.array_equal(o['connected'][i], g['obs_connected'][t])
        np.testing.assert_allclose(o['dr'][i], g['obs_dr'][t], rtol=1e-12)
        np.testing.assert_allclose(o['utility'][i], g['obs_utility'][t][0], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(o['ues_at_bs'][i], g['obs_ues_at_bs'][t], rtol=1e-12)
        np.testing.assert_allclose(o['util_at_bs'][i], g['obs_util_at_bs'][t], rtol=1e-9, atol=1e-12)
    check_obs(0)
    for t in range(len(g['actions'])):
        a = np.zeros(U, np.int32)
        a[env.time() % U] = g['actions'][t]
        env.step(a)
        check_obs(t + 1)
        assert env.reward_before()[(env.time() - 1) % U] == pytest.approx(float(g['reward'][t]), rel=1e-9, abs=1e-12)
        assert np.array_equal(env.state()['pos'], g['pos'][t]) and np.array_equal(env.state()['conn'], g['conn'][t])
