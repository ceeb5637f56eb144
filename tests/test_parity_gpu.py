"""GPU parity tests: the HIP path (through the C ABI) against (1) the golden fixtures recorded from the
reference and (2) the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): connection masks / cell indices, FSM state and FP64 positions BIT-EXACT;
SNR / data-rate floats within 1e-5 relative.  Utility is 10*log10(rate), so a 1e-5 relative rate error is a
4.3e-5 absolute utility error: utilities and rewards on the [-20, 20] scale use ATOL_UTIL = 5e-5, observation
entries (normalised to [-1, 1]) 5e-6; obs['dr'] is checked RELATIVELY down to float32's smallest normal number
(tests/parity.py holds the constants and says why).
"""
import ctypes
import glob
import os

import numpy as np
import pytest

from tests import parity

pytestmark = pytest.mark.gpu

from tests.parity import ATOL_OBS, ATOL_UTIL, RTOL_RATE  # noqa: E402  (one place for the bars)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _entities_from_fixture(g):
    from deepcomp_amd.entities import Basestation, Map, Point, RandomWaypoint, User
    inv_sh = {0: 'resource-fair', 1: 'rate-fair', 2: 'max-cap', 3: 'proportional-fair'}
    w, h = (float(x) for x in g['cfg_map_wh_raw'])
    m = Map(w, h)
    bs = [Basestation(chr(65 + i), Point(x, y), inv_sh[int(s)]) for i, ((x, y), s) in enumerate(zip(g['cfg_bs_pos'], g['cfg_bs_sharing']))]
    vel = {-1: 'slow', -2: 'fast'}
    xy = [['random' if int(c) < 0 else int(c) for c in p] for p in g['cfg_ue_init_xy']]
    U = len(g['cfg_ue_vel'])
    pause = [int(v) for v in g['cfg_ue_pause']] if 'cfg_ue_pause' in g.files else [2] * U           # movement.py:87 defaults
    border = [int(v) for v in g['cfg_ue_border']] if 'cfg_ue_border' in g.files else [10] * U
    vnum = g['cfg_ue_vel_num'] if 'cfg_ue_vel_num' in g.files else [-1.0] * U           # velocities that are no integers (movement.py:116-117)
    ues = [User(str(i + 1), m, xy[i][0], xy[i][1],
                RandomWaypoint(m, float(vnum[i]) if vnum[i] >= 0 else vel.get(int(v), int(v)), pause_duration=pause[i], border_buffer=border[i]),
                util_func='log' if int(u) == 0 else 'step', dr_req=float(r))
           for i, (v, u, r) in enumerate(zip(g['cfg_ue_vel'], g['cfg_ue_util'], g['cfg_ue_dr_req']))]
    return m, bs, ues


def _core_from_fixture(g, env_seeds=None, num_envs=1):
    from deepcomp_amd.env import BatchedMobileEnv
    m, bs, ues = _entities_from_fixture(g)
    kind = 'central' if int(g['cfg_kind']) == 0 else 'multi'
    reward = {0: 'avg', 1: 'sum', 2: 'min'}[int(g['cfg_reward'])]
    return BatchedMobileEnv(m, bs, ues, kind, num_envs=num_envs, seed=int(g['cfg_seed']), episode_length=int(g['cfg_eps_len']),
                            reward=reward, rand_episodes=bool(g['cfg_rand_episodes']), rng='reference', env_seeds=env_seeds,
                            tape_depth=64)


def _compare(core, g, prefix, i, e=0, with_reward=False):
    st = core.state_host()
    U, B = core.U, core.B
    for k in ('pos', 'wp', 'vel'):
        assert np.array_equal(st[k][e], g[f'{prefix}_{k}'][i]), f'{prefix}[{i}] {k} not bit-exact'
    assert np.array_equal(st['pausing'][e], g[f'{prefix}_pausing'][i]), f'{prefix}[{i}] pausing'
    assert np.array_equal(st['curr_pause'][e], g[f'{prefix}_curr_pause'][i]), f'{prefix}[{i}] curr_pause'
    conn = ((st['conn'][e][:, None] >> np.arange(B, dtype=st['conn'].dtype)[None, :]) & 1).astype(np.uint8)     # (uint64 masks with > 32 stations)
    assert np.array_equal(conn, g[f'{prefix}_conn'][i]), f'{prefix}[{i}] connection mask'
    np.testing.assert_allclose(st['ewma'][e], g[f'{prefix}_ewma'][i], rtol=RTOL_RATE, atol=1e-30, err_msg=f"{prefix}[{i}] ewma")
    v = {k: t.cpu().numpy()[e] for k, t in core.obs_views().items()}
    assert np.array_equal(v['connected'].reshape(U, B), g[f'{prefix}_obs_connected'][i])
    np.testing.assert_allclose(v['dr'].reshape(U, B), g[f'{prefix}_obs_dr'][i], rtol=RTOL_RATE, atol=1e-30, err_msg=f'{prefix}[{i}] obs dr')
    np.testing.assert_allclose(v['utility'].reshape(U), g[f'{prefix}_obs_utility'][i], atol=ATOL_OBS, rtol=0)
    if core.kind == 1:
        np.testing.assert_allclose(v['ues_at_bs'], g[f'{prefix}_obs_ues_at_bs'][i], atol=1e-6, rtol=0)
        np.testing.assert_allclose(v['util_at_bs'], g[f'{prefix}_obs_util_at_bs'][i], atol=ATOL_OBS, rtol=0)
    if with_reward:
        np.testing.assert_allclose(core.ue_dr.cpu().numpy()[e], g['step_curr_dr'][i], rtol=RTOL_RATE, atol=1e-30, err_msg=f'curr_dr[{i}]')
        np.testing.assert_allclose(core.ue_utility.cpu().numpy()[e], g['step_utility'][i], atol=ATOL_UTIL, rtol=0)
        r = core.reward.cpu().numpy()[e]
        tol = ATOL_UTIL if core.kind == 1 else ATOL_OBS * (U if core.reward_agg == 'sum' else 1)
        if core.kind == 1 and core.reward_agg == 'sum':
            tol = ATOL_OBS * U
        np.testing.assert_allclose(np.atleast_1d(r), g['step_reward'][i], atol=tol, rtol=0, err_msg=f'reward[{i}]')
        assert float(core.sum_utility.cpu().numpy()[e]) == pytest.approx(float(g['step_sum_utility'][i]), abs=ATOL_UTIL * U)


# ------------------------------------------------------------------------------------ device primitives
def test_fp64_sqrt_div_fma_bit_exact(torch_cuda):
    """The movement step needs IEEE-correct FP64 sqrt / divide / fma on the device (bit-exact positions)."""
    torch = torch_cuda
    from deepcomp_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(0)
    n = 1 << 20
    x = np.concatenate([rng.uniform(0, 1e5, n // 2), rng.uniform(0, 400, n // 2) ** 2])
    y = rng.uniform(1e-3, 500, n)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    out = torch.zeros(n, dtype=torch.float64, device='cuda')
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for op, want in ((0, np.sqrt(x)), (1, x / y)):
        _lib.check(L.dcomp_selftest(op, 0, xd.data_ptr(), yd.data_ptr(), out.data_ptr(), n, s))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want), f'op {op} not correctly rounded'
    _lib.check(L.dcomp_selftest(2, 0, xd.data_ptr(), yd.data_ptr(), out.data_ptr(), n, s))
    torch.cuda.synchronize()
    import math
    got = out.cpu().numpy()
    idx = rng.integers(0, n, 2000)
    from fractions import Fraction
    for i in idx:
        want = float(Fraction(float(y[i])) * Fraction(float(y[i])) + Fraction(float(x[i] * x[i])))
        assert got[i] == want
    assert math.isfinite(got.sum())


def test_move_norm_and_unit_equals_sqrt_and_division(torch_cuda):
    """move_ue's own FP64 sequence (v_rsq_f64 / v_rcp_f64 + Newton steps, ONE reciprocal for both divisions, no range
    scaling) must give the correctly rounded np.linalg.norm and quotients of movement.py:151 on its whole domain:
    waypoint minus position on maps up to 65535 m, incl. tiny residuals next to the waypoint and exact zeros."""
    torch = torch_cuda
    from deepcomp_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(3)
    n = 1 << 21
    k = n // 8
    vx = np.concatenate([rng.uniform(-600, 600, 3 * k), rng.integers(-600, 601, k).astype(np.float64),
                         rng.uniform(-65535, 65535, k), rng.uniform(-1, 1, k) * 1e-9, np.zeros(k), rng.uniform(-3, 3, k)])
    vy = np.concatenate([rng.uniform(-600, 600, 3 * k), rng.integers(-600, 601, k).astype(np.float64),
                         rng.uniform(-65535, 65535, k), rng.uniform(1, 10, k), rng.uniform(0.5, 600, k), rng.uniform(-3, 3, k)])
    vy[(vx == 0) & (vy == 0)] = 1.0                       # the move never normalises a zero vector (snap branch)
    xd, yd = torch.from_numpy(vx).cuda(), torch.from_numpy(vy).cuda()
    out = torch.zeros(n, dtype=torch.float64, device='cuda')
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(op):
        _lib.check(L.dcomp_selftest(op, 0, xd.data_ptr(), yd.data_ptr(), out.data_ptr(), n, s))
        torch.cuda.synchronize()
        return out.cpu().numpy().copy()
    q = run(2)                                           # fma(vy, vy, vx*vx): checked exactly in the test above
    nrm = np.sqrt(q)
    assert np.array_equal(run(4), nrm), 'norm is not the correctly rounded sqrt'
    assert np.array_equal(run(5), vx / nrm), 'vx / norm is not the correctly rounded quotient'
    assert np.array_equal(run(6), vy / nrm), 'vy / norm is not the correctly rounded quotient'


@pytest.mark.parametrize('width', [2, 4, 8, 16, 32, 64])
def test_group_reductions(torch_cuda, width):
    """DPP / ds_swizzle segmented all-reduce against numpy."""
    torch = torch_cuda
    from deepcomp_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(width)
    n = 4096
    x = np.round(rng.uniform(-8, 8, n) * 4) / 4          # exactly representable, sums are exact in FP32
    xd = torch.from_numpy(x).cuda()
    out = torch.zeros(n, dtype=torch.float64, device='cuda')
    _lib.check(L.dcomp_selftest(3, width, xd.data_ptr(), None, out.data_ptr(), n, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    g = x.reshape(-1, width)
    want = (g.sum(1, keepdims=True) + 1024.0 * (g > 0).sum(1, keepdims=True) + 1048576.0 * g.min(1, keepdims=True)) * np.ones_like(g)
    assert np.array_equal(out.cpu().numpy().reshape(-1, width), want)


# ------------------------------------------------------------------------------------ golden trajectories
TRAJ = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'traj_*.npz')))


@pytest.mark.parametrize('name', TRAJ)
def test_golden_trajectory(torch_cuda, name):
    torch = torch_cuda
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    core = _core_from_fixture(g)
    episodes = int(g['cfg_episodes'])
    steps = g['actions'].shape[0] // episodes
    t = 0
    for ep in range(episodes):
        core.reset()
        torch.cuda.synchronize()
        _compare(core, g, 'reset', ep)
        for _ in range(steps):
            a = torch.from_numpy(g['actions'][t].astype(np.uint8).reshape(1, -1)).cuda()
            core.step(a)
            _compare(core, g, 'step', t, with_reward=True)
            t += 1
        core.check()


@pytest.mark.parametrize('stack', ['estack_grid32x10_multi', 'estack_grid10x5_central'])
def test_golden_env_stack(torch_cuda, stack):
    """E-axis parity: 8 envs seeded 42 + 20000*e in ONE batch vs 8 reference runs."""
    torch = torch_cuda
    gs = [np.load(os.path.join(GOLDEN, f'{stack}_e{e}.npz')) for e in range(8)]
    core = _core_from_fixture(gs[0], num_envs=8)
    assert list(core.env_seeds) == [int(g['cfg_seed']) for g in gs]
    core.reset()
    for e in range(8):
        _compare(core, gs[e], 'reset', 0, e=e)
    for t in range(gs[0]['actions'].shape[0]):
        a = torch.from_numpy(np.stack([g['actions'][t] for g in gs]).astype(np.uint8)).cuda()
        core.step(a)
        for e in range(8):
            _compare(core, gs[e], 'step', t, e=e, with_reward=True)
    core.check()


# ------------------------------------------------------------------------------------ oracle at scale (Philox)
def _oracle_batch(scn, kind, reward, E, seed, env_id_base=0):
    from oracle import oracle as orc
    vel = [s['velocity'] for s in scn.ue_specs]
    envs = []
    for e in range(E):
        o = orc.OracleEnv(int(scn.width), int(scn.height), scn.bs_pos, scn.bs_sharing, vel,
                          kind=orc.MULTI if kind == 'multi' else orc.CENTRAL, reward_agg={'avg': 0, 'sum': 1, 'min': 2}[reward],
                          pause=[s.get('pause_duration', 2) for s in scn.ue_specs],
                          border=[s.get('border_buffer', 10) for s in scn.ue_specs])
        o.set_philox(seed, env_id_base + e)
        envs.append(o)
    return orc.OracleBatch(envs)


@pytest.mark.parametrize('shape', [('multi', 32, 10, 512, 'avg'), ('central', 10, 5, 1024, 'avg'), ('multi', 3, 3, 700, 'min'),
                                   ('multi', 20, 7, 300, 'sum'), ('multi', 100, 12, 64, 'avg'), ('central', 130, 6, 40, 'min'),
                                   ('multi', 64, 9, 96, 'sum'), ('multi', 70, 5, 50, 'min'), ('central', 128, 32, 12, 'avg'),
                                   ('multi', 256, 3, 10, 'avg'), ('central', 60, 16, 33, 'sum'), ('multi', 16, 11, 64, 'avg'),
                                   ('central', 8, 27, 32, 'avg'), ('multi', 5, 1, 128, 'min'), ('multi', 128, 29, 6, 'sum'),
                                   ('multi', 128, 32, 8, 'avg'), ('multi', 100, 32, 5, 'min'), ('multi', 64, 25, 8, 'avg'),
                                   ('multi', 200, 26, 3, 'avg'), ('central', 128, 32, 4, 'min'), ('central', 256, 31, 2, 'sum'),
                                   ('central', 1, 1, 64, 'avg'), ('multi', 1, 2, 64, 'sum'), ('multi', 2, 32, 16, 'min'),
                                   ('multi', 40, 6, 24, 'avg', 'max-cap'), ('multi', 128, 32, 4, 'avg', 'proportional-fair'),
                                   ('multi', 64, 26, 6, 'sum', 'rate-fair'), ('central', 100, 30, 3, 'avg', 'max-cap'),
                                   ('multi', 32, 10, 64, 'min', 'resource-fair'), ('multi', 128, 22, 6, 'avg'), ('central', 70, 21, 5, 'sum'),
                                   ('multi', 64, 24, 8, 'min', 'rate-fair')])
def test_oracle_parity_philox(torch_cuda, shape):
    """HIP path vs CPU oracle, same Philox draws, random actions, 60 steps incl. one mid-run reset."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    kind, U, B, E, reward = shape[:5]
    sharing = shape[5] if len(shape) > 5 else 'mixed'
    scn = scenarios.grid_map(B, sharing).with_ues(num_static=U // 8, num_slow=U - U // 8 - U // 4, num_fast=U // 4)
    m, bs, ues = build_from_scenario(scn)
    core = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=1234, reward=reward, rng='philox', rand_episodes=True, env_id_base=5)
    ob = _oracle_batch(scn, kind, reward, E, 1234, env_id_base=5)
    rng = np.random.default_rng(7)

    def cmp(obs_o, rew_o, conn_o, pos_o):
        # masks / positions bit-exact; per-UE rate, EWMA and obs.dr 1e-5 RELATIVE against the oracle's FP64 values (tests/parity.py)
        parity.assert_step(core, ob, obs_o, rew_o, conn_o, pos_o, kind, reward)

    core.reset()
    cmp(ob.reset(), None, None, None)
    for t in range(60):
        if t == 35:
            for i, o in enumerate(ob.envs):
                o.set_episode(1)
            core.reset()
            cmp(ob.reset(), None, None, None)
        a = rng.integers(0, B + 1, size=(E, U)).astype(np.uint8)
        a[rng.random((E, U)) < 0.5] = 0
        core.step(torch.from_numpy(a).cuda())
        cmp(*ob.step(a))
    core.check()


TIGHT_SHAPES = [('multi', 10, 5, 333, 'avg'), ('central', 10, 5, 1000, 'avg'), ('multi', 20, 10, 77, 'sum'), ('multi', 5, 3, 700, 'min'),
                ('central', 12, 9, 50, 'min'), ('multi', 17, 6, 41, 'avg'), ('multi', 7, 4, 129, 'avg', 'proportional-fair'),
                ('central', 5, 32, 77, 'sum'), ('multi', 24, 11, 30, 'min', 'rate-fair'), ('multi', 31, 2, 19, 'avg'),
                ('central', 6, 7, 300, 'avg', 'resource-fair')]


@pytest.mark.parametrize('shape', TIGHT_SHAPES)
def test_tight_packing_oracle_parity(torch_cuda, shape, monkeypatch):
    """UE lists whose length is not a power of two, packed tightly (U lanes per env, 64 // U envs per wavefront, segmented
    ds_bpermute reductions) -- forced on with DCOMP_TIGHT=1 (dcomp_create picks it by itself only for throughput-bound
    batches): the same oracle-parity run as test_oracle_parity_philox, batch sizes that leave partial wavefronts; and bit-identical
    positions / masks against the padded kernel."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    kind, U, B, E, reward = shape[:5]
    sharing = shape[5] if len(shape) > 5 else 'mixed'
    monkeypatch.setenv('DCOMP_TIGHT', '1')
    test_oracle_parity_philox(torch, shape)
    scn = scenarios.grid_map(B, sharing).with_ues(num_static=U // 8, num_slow=U - U // 8 - U // 4, num_fast=U // 4)
    m, bs, ues = build_from_scenario(scn)
    tight = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=5, reward=reward, rng='philox')
    monkeypatch.setenv('DCOMP_TIGHT', '0')
    padded = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=5, reward=reward, rng='philox')
    assert tight.lanes_per_env == U and padded.lanes_per_env == 1 << (U - 1).bit_length()
    tight.reset(); padded.reset()
    g = torch.Generator(device='cuda').manual_seed(9)
    for t in range(25):
        a = torch.randint(0, B + 1, (E, U), generator=g, device='cuda', dtype=torch.uint8)
        tight.step(a); padded.step(a)
        assert torch.equal(tight.pos, padded.pos) and torch.equal(tight.conn, padded.conn) and torch.equal(tight.mv, padded.mv), f'step {t}'
        torch.testing.assert_close(tight.obs, padded.obs, rtol=2e-6, atol=2e-6)          # sums in scan order vs butterfly order
        torch.testing.assert_close(tight.reward, padded.reward, rtol=0, atol=2e-5 * (U if reward == 'sum' else 1))
    tight.check(); padded.check()


@pytest.mark.parametrize('kind,U,B,E', [('multi', 32, 10, 256), ('central', 10, 5, 300), ('multi', 128, 32, 6), ('multi', 70, 9, 20)])
def test_movement_parameters_philox(torch_cuda, kind, U, B, E):
    """RandomWaypoint(pause_duration, border_buffer) per UE away from the defaults 2 / 10 (movement.py:87-104; round 1 refused
    them) and fixed velocities that are no integers in 0..255 (round 2 refused them): counter-based draws against the oracle, 90
    steps incl. a reset, through step() and through the fused rollout.
    (The reference-run fixtures traj_*pause_border* / traj_*velocity_numbers* pin the same parameters in tape mode.)"""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    scn = scenarios.grid_map(B, 'mixed').with_ues(num_slow=U - U // 2, num_fast=U // 2)
    for i, spec in enumerate(scn.ue_specs):
        spec['pause_duration'], spec['border_buffer'] = (0, 1, 2, 5, 9, 30, 127)[i % 7], (1, 10, 25, 49, 3)[i % 5]
        if i % 4 == 3:          # movement.py:116-117: a fixed velocity is any number -- not an integer, or beyond the movement word's 8 bits
            spec['velocity'] = (2.5, 0.3, 7.125, 300.0, 11.7, 1e-3)[(i // 4) % 6]
    m, bs, ues = build_from_scenario(scn)
    core = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=99, rng='philox', rand_episodes=True, episode_length=45)
    roll = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=99, rng='philox', rand_episodes=True, episode_length=45)
    ob = _oracle_batch(scn, kind, 'avg', E, 99)
    rng = np.random.default_rng(3)
    acts = rng.integers(0, B + 1, size=(90, E, U)).astype(np.uint8)
    dev = torch.from_numpy(acts).cuda()
    core.reset(); roll.reset(); ob.reset()
    for t in range(90):
        if t == 45:
            for o in ob.envs:
                o.set_episode(1)
            core.reset(); ob.reset()
        core.step(dev[t])
        o_obs, o_rew, o_conn, o_pos = ob.step(acts[t])
        st = core.state_host()
        assert np.array_equal(st['pos'], o_pos) and np.array_equal(st['conn'], o_conn), f'step {t}'
    parity.assert_obs(core.obs.cpu().numpy(), o_obs, kind, U, B, msg='after 90 steps')
    assert int(st['curr_pause'].max()) > 3                      # pauses longer than the default were exercised
    roll.rollout(dev, horizon=45)
    assert torch.equal(roll.pos, core.pos) and torch.equal(roll.mv, core.mv) and torch.equal(roll.obs, core.obs)
    core.check(); roll.check()


def test_random_configurations_slice(torch_cuda):
    """A fixed-seed slice of tools/fuzz_parity.py: random shapes, BS layouts, sharing models (incl. max-cap), utilities,
    velocities, start positions (incl. UEs parked ON a BS), rewards and agent kinds against the oracle.  (The full
    fuzzer found the max-cap near-tie case: squared distances one ulp apart have the same FP64 rate in the reference.)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), '..', 'tools'))
    import fuzz_parity
    rng = np.random.default_rng(0)
    for i in range(40):
        c = fuzz_parity.random_case(rng)
        try:
            fuzz_parity.run_case(c, torch_cuda)
        except AssertionError as ex:
            raise AssertionError(f'case {i}: {fuzz_parity.describe(c)}\n{ex}') from None


def test_max_cap_near_tie_goes_to_the_oldest_connection(torch_cuda):
    """station.py:183-187 compares bw*log2(1+snr) in FP64: two UEs whose squared distances to the BS differ in the last
    bits (mirrored positions reached by FP64 movement) have the SAME rate, and the first in connection order is served.
    The two configurations the fuzzer found this with, frozen in tests/golden/fuzz_maxcap_near_ties.json (inputs only;
    the oracle is the checker)."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), '..', 'tools'))
    import fuzz_parity
    for spec in json.load(open(os.path.join(GOLDEN, 'fuzz_maxcap_near_ties.json'))):
        assert 'max-cap' in spec['sh']
        fuzz_parity.run_case(fuzz_parity.build_case(spec), torch_cuda)


def test_max_cap_near_ties_the_reference_confirmed(torch_cuda):
    """The 28 configurations in which a max-cap BS sees its two closest UEs at squared distances a few ulps apart AND the
    reference itself was run against the oracle on them (tests/golden/fuzz_oracle_vs_reference.py --near-ties 28 --seed 0
    --dump ...: 28 / 28 agree; inputs only are stored): the HIP path must agree with the oracle on every one."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), '..', 'tools'))
    import fuzz_parity
    specs = json.load(open(os.path.join(GOLDEN, 'fuzz_maxcap_near_ties_reference_confirmed.json')))
    assert len(specs) == 28
    for i, spec in enumerate(specs):
        assert 'max-cap' in spec['sh'] and spec['U'] >= 30
        try:
            fuzz_parity.run_case(fuzz_parity.build_case(spec), torch_cuda)
        except AssertionError as ex:
            raise AssertionError(f'near-tie configuration {i}: {ex}') from None


@pytest.mark.parametrize('agent_name', ['fullcomp', '3gpp', 'dynamic', 'static'])
def test_heuristic_driven_rollout_matches_oracle(torch_cuda, agent_name):
    """Policy in the loop, everything on the device: a reference heuristic (dcomp_heuristic_actions via agents.py) reads the
    kernel's observation tensor and its actions drive the next step; the oracle is stepped with the same
    actions.  Sticky policies build up many simultaneous connections per BS (unlike random actions)."""
    torch = torch_cuda
    from deepcomp_amd import agents, scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    E, U, B = 256, 32, 10
    scn = scenarios.grid_map(B, 'mixed').with_ues(num_static=4, num_slow=20, num_fast=8)
    m, bs, ues = build_from_scenario(scn)
    core = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=77, rng='philox')
    ob = _oracle_batch(scn, 'multi', 'avg', E, 77)
    agent = {'fullcomp': agents.FullCoMP(), '3gpp': agents.Heuristic3GPP(), 'dynamic': agents.DynamicSelection(0.3),
             'static': agents.StaticClustering(3, bs, seed=5, device='cuda')}[agent_name]
    core.reset()
    ob.reset()
    total_conn = 0
    for t in range(50):
        act = agent.act(core)                        # HIP policy kernel on the packed observation tensor
        core.step(act)
        o_obs, o_rew, o_conn, o_pos = ob.step(act.cpu().numpy())
        st = core.state_host()
        assert np.array_equal(st['conn'], o_conn) and np.array_equal(st['pos'], o_pos)
        parity.assert_obs(core.obs.cpu().numpy(), o_obs, 'multi', core.U, core.B, msg=f'{agent_name} step {t}')
        np.testing.assert_allclose(core.reward.cpu().numpy(), o_rew, atol=ATOL_UTIL, rtol=0)
        total_conn += int(np.unpackbits(o_conn.view(np.uint8)).sum())
    core.check()
    assert total_conn > E * U * 10          # the policy really holds connections


# ------------------------------------------------------------------------------------ UE arrival / departure
DYN = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'dyn_*.npz')) + glob.glob(os.path.join(GOLDEN, 'reseeddyn_*.npz')))


def _dyn_kwargs(g):
    arr = {int(t): int(n) for t, n in zip(g['cfg_arrival_t'], g['cfg_arrival_n'])} or None
    interval = int(g['cfg_new_ue_interval'])
    return dict(ue_arrival=arr, new_ue_interval=interval if interval > 0 else None, max_ues=int(g['cfg_max_ues']))


@pytest.mark.parametrize('via_rollout', [False, True])
@pytest.mark.parametrize('name', DYN)
def test_golden_dynamic_ue_trajectory(torch_cuda, name, via_rollout):
    """UE arrival / departure (base.py:433-443, 592-618) against reference-run fixtures: slot order, ids, masks,
    FP64 positions exact (incl. the reference's reseed-by-list-position behaviour across episodes).  reseeddyn_*: with
    MobileEnv.seed() calls in the middle of episodes (base.py:132-143; listed UEs incl. arrived ones re-seeded by position, the
    departure / arrival-point generators restart)."""
    torch = torch_cuda
    from deepcomp_amd.entities import Basestation, Map, Point, RandomWaypoint, User
    from deepcomp_amd.env import BatchedMobileEnv
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    inv_sh = {0: 'resource-fair', 1: 'rate-fair', 2: 'max-cap', 3: 'proportional-fair'}
    w, h = (float(x) for x in g['cfg_map_wh_raw'])
    m = Map(w, h)
    bs = [Basestation(chr(65 + i), Point(x, y), inv_sh[int(s)]) for i, ((x, y), s) in enumerate(zip(g['cfg_bs_pos'], g['cfg_bs_sharing']))]
    vel = {-1: 'slow', -2: 'fast'}
    ues = [User(str(i + 1), m, 'random', 'random', RandomWaypoint(m, vel.get(int(v), int(v)))) for i, v in enumerate(g['cfg_ue_vel'])]
    kind = 'central' if int(g['cfg_kind']) == 0 else 'multi'
    core = BatchedMobileEnv(m, bs, ues, kind, num_envs=1, seed=int(g['cfg_seed']), episode_length=int(g['cfg_eps_len']),
                            reward={0: 'avg', 1: 'sum', 2: 'min'}[int(g['cfg_reward'])], rand_episodes=bool(g['cfg_rand_episodes']),
                            rng='reference', tape_depth=48, **_dyn_kwargs(g))
    M, B = core.U, core.B
    assert M == int(g['cfg_max_ues'])

    def cmp(prefix, i, with_reward):
        st = core.state_host()
        n = int(g[f'{prefix}_num_ue'][i])
        assert core.num_ue == n
        assert np.array_equal(st['uid'][0], g[f'{prefix}_ue_ids'][i]), f'{prefix}[{i}] ids'
        for k in ('pos', 'wp', 'vel'):
            assert np.array_equal(st[k][0], g[f'{prefix}_{k}'][i]), f'{prefix}[{i}] {k} not bit-exact'
        assert np.array_equal(st['pausing'][0], g[f'{prefix}_pausing'][i]) and np.array_equal(st['curr_pause'][0], g[f'{prefix}_curr_pause'][i])
        conn = ((st['conn'][0][:, None] >> np.arange(B, dtype=st['conn'].dtype)[None, :]) & 1).astype(np.uint8)
        assert np.array_equal(conn, g[f'{prefix}_conn'][i]), f'{prefix}[{i}] connection mask'
        np.testing.assert_allclose(st['ewma'][0], g[f'{prefix}_ewma'][i], rtol=RTOL_RATE, atol=1e-30)
        v = {k: t.cpu().numpy()[0] for k, t in core.obs_views().items()}
        assert np.array_equal(v['connected'].reshape(M, B), g[f'{prefix}_obs_connected'][i])
        np.testing.assert_allclose(v['dr'].reshape(M, B), g[f'{prefix}_obs_dr'][i], rtol=RTOL_RATE, atol=1e-30)
        np.testing.assert_allclose(v['utility'].reshape(M), g[f'{prefix}_obs_utility'][i], atol=ATOL_OBS, rtol=0)
        if kind == 'multi':
            np.testing.assert_allclose(v['ues_at_bs'], g[f'{prefix}_obs_ues_at_bs'][i], atol=1e-6, rtol=0)
            np.testing.assert_allclose(v['util_at_bs'], g[f'{prefix}_obs_util_at_bs'][i], atol=ATOL_OBS, rtol=0)
        if with_reward:
            np.testing.assert_allclose(core.ue_dr.cpu().numpy()[0], g['step_curr_dr'][i], rtol=RTOL_RATE, atol=1e-30)
            np.testing.assert_allclose(core.ue_utility.cpu().numpy()[0], g['step_utility'][i], atol=ATOL_UTIL, rtol=0)
            r = np.atleast_1d(core.reward.cpu().numpy()[0])
            np.testing.assert_allclose(r, g['step_reward'][i], atol=ATOL_UTIL if kind == 'multi' else ATOL_OBS, rtol=0)
            assert float(core.sum_utility.cpu().numpy()[0]) == pytest.approx(float(g['step_sum_utility'][i]), abs=ATOL_UTIL * M)

    seed_at = {int(t_): int(s_) for t_, s_ in g['cfg_seed_at']} if 'cfg_seed_at' in g.files else {}
    t = 0
    for ep in range(int(g['cfg_episodes'])):
        core.reset()
        cmp('reset', ep, False)
        if via_rollout:                                  # the same episode through rollout()'s event feed, in fragments of 1-7 steps
            L, frag = int(g['cfg_eps_len']), 1
            left = L
            while left:
                if t in seed_at:
                    core.seed(seed_at[t], immediate=True)
                n = min([frag, left] + [ts - t for ts in seed_at if ts > t])          # a fragment ends where the caller seeds
                acts = torch.from_numpy(g['actions'][t:t + n].astype(np.uint8).reshape(n, 1, -1)).cuda()
                core.rollout(acts)
                t += n; left -= n
                cmp('step', t - 1, True)
                frag = frag % 7 + 2
            core.check()
            continue
        for _ in range(int(g['cfg_eps_len'])):
            if t in seed_at:
                core.seed(seed_at[t], immediate=True)
            core.step(torch.from_numpy(g['actions'][t].astype(np.uint8).reshape(1, -1)).cuda())
            cmp('step', t, True)
            t += 1
        core.check()


@pytest.mark.parametrize('kind,reward', [('multi', 'avg'), ('central', 'avg'), ('multi', 'min'), ('central', 'sum')])
def test_oracle_parity_dynamic_philox(torch_cuda, kind, reward):
    """UE arrival / departure at scale: Philox-keyed departures (which UE leaves differs per env) and border points,
    identical mapping in kernel and oracle; masks / positions / ids exact."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    from oracle import oracle as orc
    E, U0, B, L = 333, 6, 7, 60
    arrival = {2: 3, 5: -2, 9: 4, 14: -3, 20: 2, 21: 2, 30: -4, 41: 5, 50: -6}
    scn = scenarios.large_map('mixed').with_ues(num_static=1, num_slow=3, num_fast=2)
    scn.ue_specs[0]['velocity'], scn.ue_specs[5]['velocity'] = 2.5, 11.7          # fixed velocities that are no integers (movement.py:116-117)
    m, bs, ues = build_from_scenario(scn)
    core = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=5, episode_length=L, reward=reward, rng='philox', rand_episodes=True,
                            ue_arrival=arrival)
    M = core.U
    sched = orc.arrival_schedule(L, arrival)
    oenvs = []
    for e in range(E):
        o = orc.OracleEnv(int(scn.width), int(scn.height), scn.bs_pos, scn.bs_sharing, [s['velocity'] for s in scn.ue_specs],
                          kind=orc.MULTI if kind == 'multi' else orc.CENTRAL, reward_agg={'avg': 0, 'sum': 1, 'min': 2}[reward], max_ues=M)
        o.set_philox(5, e)
        oenvs.append(o)
    ob = orc.OracleBatch(oenvs)
    rng = np.random.default_rng(3)
    for ep in range(2):
        for o in oenvs:
            o.set_episode(ep)
        core.reset()
        want = ob.reset()
        parity.assert_obs(core.obs.cpu().numpy(), want, kind, M, B, msg=f'episode {ep} reset')
        for t in range(L):
            a = rng.integers(0, B + 1, size=(E, M)).astype(np.uint8)
            a[rng.random((E, M)) < 0.5] = 0
            n_rem, n_add = sched[t]
            if n_rem or n_add:
                for o in oenvs:
                    o.set_event_counts(n_rem, n_add)
            core.step(torch.from_numpy(a).cuda())
            o_obs, o_rew, o_conn, o_pos = ob.step(a)
            st = core.state_host()
            assert core.num_ue == oenvs[0].num_ue()
            assert np.array_equal(st['uid'], np.stack([o.uids() for o in oenvs])), f'step {t}: UE ids differ'
            assert np.array_equal(st['conn'], o_conn) and np.array_equal(st['pos'], o_pos), f'step {t}'
            n0 = core.num_ue
            assert np.array_equal(st['vel'][0][:n0], oenvs[0].state()['vel'][:n0]), f'step {t}: velocities (slots shift when UEs leave)'
            parity.assert_obs(core.obs.cpu().numpy(), o_obs, kind, M, B, msg=f'episode {ep} step {t}')
            tol = (ATOL_UTIL if kind == 'multi' else ATOL_OBS) * (M if reward == 'sum' else 1)
            np.testing.assert_allclose(core.reward.cpu().numpy(), o_rew, atol=tol, rtol=0)
    core.check()


def test_single_agent_env_matches_reference(torch_cuda):
    """'--agent single' drop-in (RelNormEnv): fixtures recorded from the reference's RelNormEnv."""
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import make_env_config
    from deepcomp_amd.env import RelNormEnv, get_env_class
    assert get_env_class('single') is RelNormEnv
    for name, scn in [('single_custom3x4_s42', scenarios.custom_map('mixed').with_ues(num_slow=2, num_fast=1)),
                      ('single_small2x2_s43', scenarios.small_map('mixed').with_ues(num_slow=2))]:
        g = np.load(os.path.join(GOLDEN, name + '.npz'))
        env = RelNormEnv(make_env_config(scn, seed=int(g['cfg_seed']), episode_length=60))
        obs = env.reset()
        assert obs['connected'] == g['obs_connected'][0].astype(int).tolist()
        np.testing.assert_allclose(obs['dr'], g['obs_dr'][0], rtol=RTOL_RATE)
        for t, a in enumerate(g['actions']):
            obs, rew, done, info = env.step(int(a))
            assert done is None and isinstance(rew, float)
            assert rew == pytest.approx(float(g['reward'][t]), abs=ATOL_OBS)
            assert obs['connected'] == g['obs_connected'][t + 1].astype(int).tolist()
            np.testing.assert_allclose(obs['dr'], g['obs_dr'][t + 1], rtol=RTOL_RATE, atol=1e-30)
            np.testing.assert_allclose(obs['utility'], g['obs_utility'][t + 1], atol=ATOL_OBS)
            np.testing.assert_allclose(obs['ues_at_bs'], g['obs_ues_at_bs'][t + 1], atol=1e-6)
            np.testing.assert_allclose(obs['util_at_bs'], g['obs_util_at_bs'][t + 1], atol=ATOL_OBS)
        with pytest.raises(AssertionError):
            env.step(env.num_bs + 1)                                               # base.py:238


def test_rllib_style_adapters(torch_cuda):
    """VectorEnv / BaseEnv protocol adapters over one batch == E independent single-env drop-in instances
    (same seeds: env e of the batch is seeded seed + 20000*e)."""
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import make_env_config
    from deepcomp_amd.env import CentralRelNormEnv, MultiAgentMobileEnv
    from deepcomp_amd.rllib_adapter import CentralVectorEnv, MultiAgentBaseEnv
    E, T = 3, 12
    rng = np.random.default_rng(2)
    # central
    scn = lambda: scenarios.medium_map('mixed').with_ues(num_slow=2, num_fast=1)      # noqa: E731
    vec = CentralVectorEnv(make_env_config(scn(), seed=7, num_envs=E, rng='reference'))
    singles = [CentralRelNormEnv(make_env_config(scn(), seed=7 + 20000 * e)) for e in range(E)]
    obs = vec.vector_reset()
    for e, s in enumerate(singles):
        o = s.reset()
        assert obs[e]['connected'].tolist() == o['connected'] and np.allclose(obs[e]['dr'], o['dr'], atol=1e-6)
    for t in range(T):
        acts = rng.integers(0, 4, size=(E, 3))
        obs, rew, dones, infos = vec.vector_step([a.tolist() for a in acts])
        assert dones == [False] * E and infos[0]['time'] == t + 1
        for e, s in enumerate(singles):
            o, r, _, _ = s.step(acts[e].tolist())
            assert obs[e]['connected'].tolist() == o['connected'] and rew[e] == pytest.approx(r, abs=1e-6)
            assert np.allclose(obs[e]['utility'], o['utility'], atol=1e-6)
    assert vec.observation_space.spaces['dr'].shape == (9,) and vec.num_envs == E
    # multi-agent
    scn = lambda: scenarios.custom_map('mixed').with_ues(num_slow=3)                  # noqa: E731
    base = MultiAgentBaseEnv(make_env_config(scn(), seed=11, num_envs=E, rng='reference'))
    singles = [MultiAgentMobileEnv(make_env_config(scn(), seed=11 + 20000 * e)) for e in range(E)]
    obs, rew, dones, infos, _ = base.poll()
    for e, s in enumerate(singles):
        o = s.reset()
        assert sorted(obs[e].keys()) == ['1', '2', '3'] and obs[e]['2']['connected'].tolist() == o['2']['connected']
    for t in range(T):
        acts = {e: {str(i + 1): int(rng.integers(0, 5)) for i in range(3)} for e in range(E)}
        base.send_actions(acts)
        obs, rew, dones, infos, _ = base.poll()
        for e, s in enumerate(singles):
            o, r, _, _ = s.step(acts[e])
            for aid in ('1', '2', '3'):
                assert obs[e][aid]['connected'].tolist() == o[aid]['connected']
                assert rew[e][aid] == pytest.approx(r[aid], abs=1e-5)
                assert np.allclose(obs[e][aid]['util_at_bs'], o[aid]['util_at_bs'], atol=1e-6)
        assert dones[0] == {'__all__': False}


def test_reference_surface_with_ue_arrival(torch_cuda):
    """Drop-in class with env_config['ue_arrival']: observation / reward dicts follow the UE ids of the moment."""
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import make_env_config
    from deepcomp_amd.env import CentralRelNormEnv, MultiAgentMobileEnv
    g = np.load(os.path.join(GOLDEN, 'dyn_custom_multi_updown_s42.npz'))
    arr = {str(int(t)): int(n) for t, n in zip(g['cfg_arrival_t'], g['cfg_arrival_n'])}
    env = MultiAgentMobileEnv(make_env_config(scenarios.custom_map('mixed').with_ues(num_slow=2), seed=42, episode_length=40,
                                              ue_arrival=arr))
    assert env.max_ues == int(g['cfg_max_ues'])
    obs = env.reset()
    assert sorted(obs.keys()) == ['1', '2']
    for t in range(40):
        ids = [ue.id for ue in env.ue_list]
        obs, rew, done, info = env.step({uid: int(g['actions'][t][i]) for i, uid in enumerate(ids)})
        want_ids = [str(x) for x in g['step_ue_ids'][t][:int(g['step_num_ue'][t])]]
        assert [ue.id for ue in env.ue_list] == want_ids and sorted(obs.keys()) == sorted(want_ids) == sorted(rew.keys())
        np.testing.assert_allclose([rew[i] for i in want_ids], g['step_reward'][t][:len(want_ids)], atol=ATOL_UTIL)
        assert [obs[i]['connected'] for i in want_ids] == g['step_obs_connected'][t][:len(want_ids)].astype(int).tolist()
    g = np.load(os.path.join(GOLDEN, 'dyn_custom_central_interval_s43.npz'))
    env = CentralRelNormEnv(make_env_config(scenarios.custom_map('mixed').with_ues(num_slow=1, num_fast=1), seed=43, episode_length=30,
                                            new_ue_interval=7))
    assert env.max_ues == 6 and env.action_space.shape == (6,)
    obs = env.reset()
    for t in range(30):
        obs, rew, done, info = env.step([int(x) for x in g['actions'][t]])
        assert len(obs['connected']) == 6 * 4 and env.num_ue == int(g['step_num_ue'][t])
        assert obs['connected'] == g['step_obs_connected'][t].astype(int).reshape(-1).tolist()
        assert rew == pytest.approx(float(g['step_reward'][t][0]), abs=ATOL_OBS)


# ------------------------------------------------------------------------------------ full-size properties
def test_full_size_properties(torch_cuda):
    """BASELINE config 3 (65 536 envs x 32 UE x 10 BS, multi-agent): size-independent invariants."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    E, U, B = 65536, 32, 10
    scn = scenarios.grid_map(B, 'mixed').with_ues(num_slow=U)
    m, bs, ues = build_from_scenario(scn)

    def run(E_, base, steps, seed=42):
        core = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E_, seed=seed, rng='philox', env_id_base=base)
        g = torch.Generator(device='cuda').manual_seed(3)
        acts = torch.randint(0, B + 1, (steps, E, U), generator=g, device='cuda', dtype=torch.uint8)
        core.reset()
        for t in range(steps):
            core.step(acts[t, base:base + E_].contiguous())
        core.check()
        return core

    full = run(E, 0, 25)
    obs = full.obs_views()
    conn_bits = full.conn.view(E, U)
    # (1) observation ranges of the reference's Box spaces (variants.py:255-268)
    assert float(obs['dr'].min()) >= 0 and float(obs['dr'].max()) <= 1.0
    assert float(obs['dr'].max(dim=-1).values.min()) == 1.0            # every row is normalised by its own max
    assert float(obs['utility'].min()) >= -1 and float(obs['utility'].max()) <= 1
    assert float(obs['util_at_bs'].min()) >= -1 and float(obs['util_at_bs'].max()) <= 1
    # (2) connected flags == connection bit mask; ues_at_bs == per-BS popcount / U
    bits = ((conn_bits.unsqueeze(-1) >> torch.arange(B, device='cuda')) & 1).float()
    assert torch.equal(bits, obs['connected'])
    assert torch.allclose(obs['ues_at_bs'], (bits.sum(1, keepdim=True) / U).expand(-1, U, -1), atol=1e-6)
    # (3) a connection implies in range: distance to that BS below the connect threshold (station.py:222-226)
    pos = full.pos.view(E, U, 2)
    bsxy = torch.tensor(scn.bs_pos, dtype=torch.float64, device='cuda')
    d = torch.linalg.norm(pos.unsqueeze(2) - bsxy.view(1, 1, B, 2), dim=-1)
    assert bool(((bits == 0) | (d < 68.92488308058013 + 1e-9)).all())
    # (4) UEs never leave the map (movement.py:165-166)
    assert float(pos.min()) >= 0 and float(pos[..., 0].max()) <= int(scn.width) and float(pos[..., 1].max()) <= int(scn.height)
    # (5) determinism + shard invariance: envs [4096, 4096+512) recomputed alone (env_id_base) are bit-identical
    part = run(512, 4096, 25)
    sl = slice(4096 * U, (4096 + 512) * U)
    assert torch.equal(part.pos, full.pos[sl]) and torch.equal(part.conn, full.conn[sl]) and torch.equal(part.mv, full.mv[sl])
    assert torch.equal(part.obs, full.obs[4096:4096 + 512])
    assert torch.equal(part.reward, full.reward[4096:4096 + 512])
    # (6) sum_utility == sum over UEs of utility; checksum of per-env sums == total
    assert torch.allclose(full.sum_utility, full.ue_utility.sum(1), atol=1e-3)


def test_full_size_oracle_parity(torch_cuda):
    """BASELINE config 3 at full size (65 536 x 32 x 10, multi-agent, mixed sharing): 6 steps of the HIP path against
    the CPU oracle on all 2 097 152 UEs -- masks and FP64 positions bit-exact; per-UE data rate, EWMA and the relative-SNR
    observation block within 1e-5 RELATIVE (north_star's bar), the utility-scaled blocks within 1e-5 absolute on [-1, 1]."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    E, U, B = 65536, 32, 10
    scn = scenarios.grid_map(B, 'mixed').with_ues(num_slow=U)
    m, bs, ues = build_from_scenario(scn)
    core = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=42, rng='philox')
    ob = _oracle_batch(scn, 'multi', 'avg', E, 42)
    rng = np.random.default_rng(11)
    core.reset()
    parity.assert_step(core, ob, ob.reset(), None, None, None, 'multi', msg='reset')
    for t in range(6):
        a = rng.integers(0, B + 1, size=(E, U)).astype(np.uint8)
        core.step(torch.from_numpy(a).cuda())
        # masks + FP64 positions bit-exact; ue_dr, ewma, obs.dr within 1e-5 RELATIVE of the oracle's FP64 values on all 2 M UEs
        parity.assert_step(core, ob, *ob.step(a), 'multi', msg=f'step {t}')
    core.check()


@pytest.mark.parametrize('kind,U,B,E', [('multi', 32, 32, 128), ('central', 10, 5, 256), ('multi', 128, 32, 16), ('multi', 12, 10, 256)])
def test_obs_dr_small_entries_are_relative(torch_cuda, kind, U, B, E):
    """obs['dr'] = snr_b / max_b snr (variants.py:276-284) on entries FAR below 1e-9: static UEs stand 0 / 1e-9 / 1e-6 / 1e-4 / 1e-2 /
    0.3 m from a station (the station's coordinate carries the offset; UE start positions are integers), which pushes their other
    entries to 1e-50 ... 1e-10; the far stations of the 32-station map add 1e-13.  tests/parity.py holds every entry that float32 can
    represent as a normal number to 1e-5 RELATIVE of the oracle's FP64 value, and what lies below to flushed / denormal."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    from oracle import oracle as orc
    offs = (0.0, 1e-9, 1e-6, 1e-4, 1e-2, 0.3)[:min(6, B)]
    n_static = len(offs)
    scn = scenarios.grid_map(B, 'mixed').with_ues(num_static=n_static, num_slow=U - n_static - U // 4, num_fast=U // 4)
    for i, off in enumerate(offs):
        x, y = scn.bs_pos[i]
        scn.ue_specs[i]['pos_x'], scn.ue_specs[i]['pos_y'] = int(x), int(y)
        scn.bs_pos[i] = (x + off, y)
    m, bs, ues = build_from_scenario(scn)
    core = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=13, rng='philox')
    init_xy = [(s['pos_x'], s['pos_y']) if s['pos_x'] != 'random' else (-1, -1) for s in scn.ue_specs]
    envs = []
    for e in range(E):
        o = orc.OracleEnv(int(scn.width), int(scn.height), scn.bs_pos, scn.bs_sharing, [s['velocity'] for s in scn.ue_specs],
                          kind=orc.MULTI if kind == 'multi' else orc.CENTRAL, init_xy=init_xy)
        o.set_philox(13, e)
        envs.append(o)
    ob = orc.OracleBatch(envs)
    rng = np.random.default_rng(4)
    core.reset()
    parity.assert_step(core, ob, ob.reset(), None, None, None, kind, msg='reset')
    seen_small = 0
    for t in range(12):
        a = rng.integers(0, B + 1, size=(E, U)).astype(np.uint8)
        core.step(torch.from_numpy(a).cuda())
        parity.assert_step(core, ob, *ob.step(a), kind, msg=f'step {t}')
        rel = ob.rates(want_dr_rel=True)['dr_rel']
        seen_small += int(((rel < 1e-9) & (rel >= parity.F32_MIN_NORMAL)).sum())
    assert seen_small > 100 * E // 16, 'the scenario no longer produces entries between 2^-126 and 1e-9'
    core.check()


@pytest.mark.parametrize('E,U,B,steps,seed_base', [(4096, 128, 32, 4, 3 * 4096), (32768, 32, 10, 5, 5 * 32768)])
def test_per_gpu_shares_oracle_parity(torch_cuda, E, U, B, steps, seed_base):
    """One GPU's share of BASELINE config 5 (4 096 x 128 UE x 32 BS: the wide kernel with its sparse pre-move pass and
    transposed per-station sums) and of config 4 (32 768 x 32 x 10), at FULL size against the CPU oracle, as the rank that
    owns global envs [seed_base, seed_base + E): masks and FP64 positions bit-exact, every observation entry and reward
    within tolerance.  Actions are biased towards in-range stations so that many UEs hold several connections at once."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    scn = scenarios.grid_map(B, 'mixed').with_ues(num_static=U // 16, num_slow=U - U // 16 - U // 4, num_fast=U // 4)
    m, bs, ues = build_from_scenario(scn)
    core = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=42, rng='philox', env_id_base=seed_base)
    ob = _oracle_batch(scn, 'multi', 'avg', E, 42, env_id_base=seed_base)
    rng = np.random.default_rng(5)
    core.reset()
    parity.assert_step(core, ob, ob.reset(), None, None, None, 'multi', msg='reset')
    bsx = np.array([q[0] for q in scn.bs_pos]); bsy = np.array([q[1] for q in scn.bs_pos])
    nconn = 0
    for t in range(steps):
        pos = core.state_host()['pos']
        d2 = (pos[:, :, 0:1] - bsx[None, None, :]) ** 2 + (pos[:, :, 1:2] - bsy[None, None, :]) ** 2
        near = np.argsort(d2, axis=2)[:, :, :3]                                     # the three closest stations
        pick = np.take_along_axis(near, rng.integers(0, 3, size=(E, U, 1)), axis=2)[:, :, 0] + 1
        a = np.where(rng.random((E, U)) < 0.7, pick, rng.integers(0, B + 1, size=(E, U))).astype(np.uint8)
        core.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, o_conn, o_pos = ob.step(a)
        parity.assert_step(core, ob, o_obs, o_rew, o_conn, o_pos, 'multi', msg=f'step {t}')
        nconn = int(np.unpackbits(o_conn.view(np.uint8)).sum())
    assert nconn > E * U // 4, 'the action bias should leave many connections in place'
    core.check()


@pytest.mark.parametrize('kind,U,B,E,reward,sharing', [('multi', 128, 32, 6, 'avg', 'mixed'), ('multi', 64, 24, 9, 'min', 'proportional-fair'),
                                                       ('central', 100, 29, 4, 'sum', 'mixed'), ('multi', 32, 10, 40, 'sum', 'rate-fair')])
def test_dense_cells_many_connections(torch_cuda, kind, U, B, E, reward, sharing):
    """Stations 25 m apart: a UE is in range of ~20 of them and, under random toggles, connected to ~10 at once -- the wide
    kernel's dense fall-backs (more than four connections per UE: no sparse post-move rate pass) and long need-sets of its
    sparse pre-move pass; every station shared by many UEs.  Against the oracle, 40 steps."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    scn = scenarios.grid_map(B, sharing, pitch=25, border=30).with_ues(num_static=U // 8, num_slow=U - U // 8 - U // 4, num_fast=U // 4)
    m, bs, ues = build_from_scenario(scn)
    core = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=31, reward=reward, rng='philox', env_id_base=2)
    ob = _oracle_batch(scn, kind, reward, E, 31, env_id_base=2)
    rng = np.random.default_rng(17)
    core.reset(); ob.reset()
    most = 0
    for t in range(40):
        a = rng.integers(0, B + 1, size=(E, U)).astype(np.uint8)
        core.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, o_conn, o_pos = ob.step(a)
        parity.assert_step(core, ob, o_obs, o_rew, o_conn, o_pos, kind, reward, msg=f'step {t}')
        most = max(most, int(np.unpackbits(o_conn.view(np.uint8).reshape(E, U, 4), axis=2).sum(axis=2).max()))
    assert most > 4, most
    core.check()


def test_long_horizon_soak(torch_cuda):
    """3 000 steps (30 episodes, resets in between) of 64 envs against the oracle: FP64 positions and masks must
    still be bit-identical at the end -- no drift, no error flag (UE outside the map, bad action)."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    E, U, B, L = 64, 32, 10, 100
    scn = scenarios.grid_map(B, 'mixed').with_ues(num_static=2, num_slow=22, num_fast=8)
    m, bs, ues = build_from_scenario(scn)
    core = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=2024, rng='philox', rand_episodes=True, episode_length=L)
    ob = _oracle_batch(scn, 'multi', 'avg', E, 2024)
    rng = np.random.default_rng(99)
    for ep in range(30):
        for o in ob.envs:
            o.set_episode(ep)
        core.reset()
        ob.reset()
        acts = rng.integers(0, B + 1, size=(L, E, U)).astype(np.uint8)
        acts[rng.random((L, E, U)) < 0.6] = 0
        dev_acts = torch.from_numpy(acts).cuda()
        for t in range(L):
            core.step(dev_acts[t])
            o_obs, o_rew, o_conn, o_pos = ob.step(acts[t])
        st = core.state_host()
        assert np.array_equal(st['pos'], o_pos) and np.array_equal(st['conn'], o_conn), f'episode {ep}'
        np.testing.assert_allclose(core.obs.cpu().numpy(), o_obs, rtol=RTOL_RATE, atol=ATOL_OBS)
        np.testing.assert_allclose(core.ewma.cpu().numpy().reshape(E, U), np.stack([o.state()['ewma'] for o in ob.envs]),
                                   rtol=RTOL_RATE, atol=1e-30)
    core.check()


def test_bad_action_flag(torch_cuda):
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    scn = scenarios.custom_map('mixed').with_ues(num_slow=4)
    m, bs, ues = build_from_scenario(scn)
    core = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=16, rng='philox')
    core.reset()
    a = torch.zeros((16, 4), dtype=torch.uint8, device='cuda')
    a[3, 2] = 9                                      # B = 4: outside the action space
    core.step(a)
    with pytest.raises(AssertionError):              # base.py:238 / central.py:61
        core.check()
    core.step(torch.zeros((16, 4), dtype=torch.uint8, device='cuda'))
    core.check()                                     # flag was cleared


def test_tape_exhaustion_is_reported(torch_cuda):
    """Reference-RNG mode with a deliberately short draw tape: the kernel flags it, check() raises (no silent reuse)."""
    torch = torch_cuda
    from deepcomp_amd import _lib, scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    scn = scenarios.small_map('mixed').with_ues(num_fast=6)
    m, bs, ues = build_from_scenario(scn)
    core = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=4, seed=1, rng='reference', tape_depth=1, episode_length=400)
    core._ensure_tape = lambda steps=1: None            # the env extends its tape on demand; without that the device flag is the net
    core.reset()
    a = torch.zeros((4, 6), dtype=torch.uint8, device='cuda')
    for _ in range(200):
        core.step(a)
    with pytest.raises(_lib.DcompError, match='tape'):
        core.check()


@pytest.mark.parametrize('rand_episodes,dyn', [(False, False), (True, False), (True, True)])
def test_tape_is_extended_for_episodes_that_outlive_it(torch_cuda, rand_episodes, dyn):
    """rng='reference' past the horizon the tape was sized for (the reference's done() is always None, --cont-train never
    resets: main.py:48-51): the env continues the SAME stdlib streams -- identical to an env whose tape was long enough from
    the start, over two episodes (so the generator states handed to the next episode line up too)."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    scn = scenarios.small_map('mixed').with_ues(num_fast=4, num_slow=2)
    m, bs, ues = build_from_scenario(scn)
    kw = dict(num_envs=3, seed=5, rng='reference', rand_episodes=rand_episodes, episode_length=20)
    if dyn:
        kw.update(ue_arrival={3: 1, 9: -2, 15: 1})
    short = BatchedMobileEnv(m, bs, ues, 'multi', tape_depth=2, **kw)
    long_ = BatchedMobileEnv(m, bs, ues, 'multi', tape_depth=200, **kw)
    g = torch.Generator(device='cuda').manual_seed(1)
    for ep in range(2):
        short.reset(); long_.reset()
        assert torch.equal(short.obs, long_.obs)
        for t in range(150 if not dyn else 19):
            if t in (4, 11):                                     # seed() on the live env (base.py:132-143): the spliced tape is extended like any other
                short.seed(900 + t + ep, immediate=True); long_.seed(900 + t + ep, immediate=True)
            a = torch.randint(0, len(bs) + 1, (3, short.U), generator=g, device='cuda', dtype=torch.uint8)
            short.step(a); long_.step(a)
        short.check(); long_.check()
        assert torch.equal(short.pos, long_.pos) and torch.equal(short.mv, long_.mv) and torch.equal(short.obs, long_.obs)
    assert short.tape_depth > 2


def test_default_tape_covers_short_pauses(torch_cuda):
    """the default tape of rng='reference' was sized for pause_duration = 2 (a redraw at most every third step).
    With pause_duration 0 / 1 and a velocity that covers the small map in one step a UE redraws every (other) step: the
    default depth and the on-demand extension now follow the smallest configured pause -- no DCOMP_FLAG_TAPE_EMPTY, and the
    same trajectory as with an amply sized tape, within one episode and across two that outlive it."""
    torch = torch_cuda
    from deepcomp_amd.entities import Basestation, Map, Point, RandomWaypoint, User
    from deepcomp_amd.env import BatchedMobileEnv
    m = Map(31, 27)
    bs = [Basestation('A', Point(10, 10), 'resource-fair'), Basestation('B', Point(22, 15), 'rate-fair')]
    ues = [User(str(i + 1), m, 'random', 'random', RandomWaypoint(m, vel, pause_duration=pd, border_buffer=bb))
           for i, (vel, pd, bb) in enumerate([(60, 0, 10), ('fast', 0, 3), (40, 1, 10), ('slow', 2, 10), (9, 0, 13)])]
    kw = dict(num_envs=5, seed=3, rng='reference', episode_length=60)
    for rand in (False, True):
        dflt = BatchedMobileEnv(m, bs, ues, 'multi', rand_episodes=rand, **kw)
        ample = BatchedMobileEnv(m, bs, ues, 'multi', rand_episodes=rand, tape_depth=400, **kw)
        assert dflt.tape_depth >= 60                     # one triple per step must fit (pause_duration 0)
        g = torch.Generator(device='cuda').manual_seed(2)
        for ep in range(2):
            dflt.reset(); ample.reset()
            for t in range(60 if ep == 0 else 150):      # the second episode outlives the default tape: extended on demand
                a = torch.randint(0, 3, (5, 5), generator=g, device='cuda', dtype=torch.uint8)
                dflt.step(a); ample.step(a)
            dflt.check(); ample.check()                  # raises on DCOMP_FLAG_TAPE_EMPTY
            assert torch.equal(dflt.pos, ample.pos) and torch.equal(dflt.mv, ample.mv) and torch.equal(dflt.obs, ample.obs)
        cursors = dflt.state_host()['cursor']
        assert int(cursors[:, 0].min()) >= 100           # the pause-0 UE on the one-step map drew (nearly) every step


def test_checkpoint_resume_continues_a_heuristic_driven_run(torch_cuda):
    """load_state_dict() left next_action holding the decision for the PRE-restore observation, so a heuristic-driven
    run resumed with one stale action.  The restored env must continue `step(env.next_action)` bit-identically."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    scn = scenarios.grid_map(7, 'mixed').with_ues(num_slow=9, num_fast=3)
    m, bs, ues = build_from_scenario(scn)

    def make():
        e = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=40, seed=8, rng='philox', rand_episodes=True, episode_length=50)
        assert e.set_policy('dynamic', epsilon=0.4)
        return e
    a = make()
    a.reset()
    for _ in range(7):
        a.step(a.next_action)
    sd = a.state_dict()
    b = make()
    b.reset()
    for _ in range(3):                                   # b is somewhere else, with another next_action
        b.step(b.next_action)
    b.load_state_dict(sd)
    assert torch.equal(a.next_action, b.next_action)
    for _ in range(9):
        a.step(a.next_action); b.step(b.next_action)
    assert torch.equal(a.pos, b.pos) and torch.equal(a.conn, b.conn) and torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward)
    a.check(); b.check()


def test_checkpoint_after_compact_stepping_restores_the_policys_own_decision(torch_cuda):
    """step_compact / step_into / rollout(out=...) never update self.obs, so a decision re-derived from the saved output
    buffer is stale.  state_dict() now carries next_action itself: a run that steps into compact records, checkpoints and resumes in
    another env continues bit-identically; checkpoints of another format say so instead of 'differently configured'."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    scn = scenarios.grid_map(6, 'mixed').with_ues(num_slow=10, num_fast=2)
    m, bs, ues = build_from_scenario(scn)

    def make():
        e = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=48, seed=21, rng='philox', rand_episodes=True, episode_length=50)
        assert e.set_policy('3gpp')
        return e
    a = make()
    a.reset()
    rec = torch.empty((48, a.compact_words), dtype=torch.int32, device='cuda')
    rew = torch.empty_like(a.reward)
    for _ in range(6):
        a.step_compact(a.next_action, rec, rew)          # self.obs still holds the reset observation
    stale = a.heuristic_actions('3gpp', obs=a.obs)
    assert not torch.equal(stale, a.next_action)         # what re-deriving from the saved output buffer would have restored
    sd = a.state_dict()
    assert sd['next_action'] is not None and torch.equal(sd['next_action'].cuda(), a.next_action)
    b = make()
    b.reset()
    b.step(b.next_action)
    b.load_state_dict(sd)
    assert torch.equal(a.next_action, b.next_action)
    ra, rb = torch.empty_like(rec), torch.empty_like(rec)
    for _ in range(8):
        a.step_compact(a.next_action, ra, rew); b.step_compact(b.next_action, rb, rew)
        assert torch.equal(ra, rb)
    assert torch.equal(a.pos, b.pos) and torch.equal(a.conn, b.conn) and torch.equal(a.ewma, b.ewma)
    a.check(); b.check()
    # formats: v2 (round 3: no 'velocity' key) is read; anything else is refused by NAME, not as "differently configured"
    old = dict(sd, config={k: v for k, v in sd['config'].items() if k != 'velocity'})
    old['config']['state_layout'] = 2
    old['next_action'] = None
    make().load_state_dict(old)
    with pytest.raises(ValueError, match='checkpoint format v1 is not supported'):
        make().load_state_dict(dict(sd, config=dict(sd['config'], state_layout=1)))
    with pytest.raises(ValueError, match='differs in: E'):
        BatchedMobileEnv(m, bs, ues, 'multi', num_envs=47, seed=21, rng='philox', rand_episodes=True, episode_length=50).load_state_dict(sd)


def test_seed_on_a_live_env(torch_cuda):
    """MobileEnv.seed (base.py:132-143) on an existing env, counter-based and tape draws: after seed(s); reset() the env is
    indistinguishable from one constructed with seed s (round 1 raised NotImplementedError for Philox envs)."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    scn = scenarios.medium_map('mixed').with_ues(num_slow=3, num_fast=2)
    m, bs, ues = build_from_scenario(scn)
    acts = torch.randint(0, len(bs) + 1, (12, 16, 5), device='cuda', dtype=torch.uint8)
    for rng, rand in (('philox', True), ('philox', False), ('reference', False), ('reference', True)):
        a = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=16, seed=11, rng=rng, rand_episodes=rand, episode_length=6)
        a.reset()
        for t in range(4):
            a.step(acts[t])
        a.seed(None)                                     # no-op (base.py:133)
        assert a.seed_value == 11
        if rng == 'philox':                              # the new key is staged until reset(): the episode in progress keeps its draws
            c = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=16, seed=11, rng=rng, rand_episodes=rand, episode_length=6)
            c.reset()
            for t in range(4):
                c.step(acts[t])
            a.seed(977)
            a.step(acts[4]); c.step(acts[4]); a.step(acts[5]); c.step(acts[5])
            assert torch.equal(a.pos, c.pos) and torch.equal(a.mv, c.mv) and torch.equal(a.obs, c.obs), (rng, rand)
        a.seed(977)
        b = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=16, seed=977, rng=rng, rand_episodes=rand, episode_length=6)
        for ep in range(2):
            a.reset(); b.reset()
            assert torch.equal(a.obs, b.obs), (rng, rand, ep)
            for t in range(6):
                a.step(acts[t + ep]); b.step(acts[t + ep])
            assert torch.equal(a.pos, b.pos) and torch.equal(a.mv, b.mv) and torch.equal(a.obs, b.obs), (rng, rand, ep)
        assert a._fingerprint() == b._fingerprint()


RESEED = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'reseed_*.npz')))


@pytest.mark.parametrize('surface', [False, True])
@pytest.mark.parametrize('name', RESEED)
def test_seed_on_a_live_env_like_the_reference(torch_cuda, name, surface):
    """MobileEnv.seed() (base.py:132-143) to the letter, against trajectories the reference itself ran with seed() calls in the
    middle of episodes and right before a reset: the running episode continues on the START of the new streams at once; reset()
    of a rand_episodes=False env re-seeds with the configured seed again (base.py:171-173), a rand_episodes=True env keeps the
    new streams.  surface: through the drop-in class (`env.seed(s)` as a caller of the reference would write it)."""
    torch = torch_cuda
    assert RESEED
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    seed_at = {int(t): int(s) for t, s in g['cfg_seed_at']}
    before_reset = {int(e): int(s) for e, s in g['cfg_seed_before_reset']}
    if surface:
        from deepcomp_amd.env import CentralRelNormEnv, MultiAgentMobileEnv
        m, bs, ues = _entities_from_fixture(g)
        cfg = {'episode_length': int(g['cfg_eps_len']), 'seed': int(g['cfg_seed']), 'map': m, 'bs_list': bs, 'ue_list': ues,
               'rand_episodes': bool(g['cfg_rand_episodes']), 'new_ue_interval': None, 'reward': 'avg', 'max_ues': None,
               'ue_arrival': None, 'log_metrics': True}
        env = (CentralRelNormEnv if int(g['cfg_kind']) == 0 else MultiAgentMobileEnv)(cfg)
        core, seed = env.core, env.seed
    else:
        core = _core_from_fixture(g)
        seed = lambda s_: core.seed(s_, immediate=True)                                   # noqa: E731
    episodes = int(g['cfg_episodes'])
    steps = g['actions'].shape[0] // episodes
    t = 0
    for ep in range(episodes):
        if ep in before_reset:
            seed(before_reset[ep])
        core.reset()
        _compare(core, g, 'reset', ep)
        for _ in range(steps):
            if t in seed_at:
                seed(seed_at[t])
            a = g['actions'][t].astype(np.uint8).reshape(1, -1)
            if surface:                      # the host-I/O mode of the E = 1 classes reads actions from its pinned buffer
                core.action_host.copy_(torch.from_numpy(a))
                core.step(core.action_host)
            else:
                core.step(torch.from_numpy(a).cuda())
            _compare(core, g, 'step', t, with_reward=True)
            t += 1
        core.check()


@pytest.mark.parametrize('dyn', [False, True])
def test_seed_before_the_first_reset_like_the_reference(torch_cuda, dyn):
    """MobileEnv.seed(s) on an env that has not been reset yet (base.py:132-143, 171-173): a rand_episodes=True env starts its
    first episode on the new streams (= an env constructed with seed s); a rand_episodes=False env re-seeds with the CONFIGURED
    seed at reset(), so the call changes nothing."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    m, bs, ues = build_from_scenario(scenarios.medium_map('mixed').with_ues(num_slow=2, num_fast=2))
    kw = dict(num_envs=3, rng='reference', episode_length=12)
    if dyn:
        kw.update(ue_arrival={2: 1, 5: -1, 7: 2})
    acts = torch.randint(0, len(bs) + 1, (12, 3, 8), device='cuda', dtype=torch.uint8)
    for rand, same_as in ((True, 977), (False, 5)):
        a = BatchedMobileEnv(m, bs, ues, 'multi', seed=5, rand_episodes=rand, **kw)
        b = BatchedMobileEnv(m, bs, ues, 'multi', seed=same_as, rand_episodes=rand, **kw)
        a.seed(977, immediate=True)
        for ep in range(2):
            a.reset(); b.reset()
            assert torch.equal(a.obs, b.obs), (rand, ep)
            for t in range(12):
                a.step(acts[t][:, :a.U].contiguous()); b.step(acts[t][:, :b.U].contiguous())
            assert torch.equal(a.pos, b.pos) and torch.equal(a.mv, b.mv) and torch.equal(a.obs, b.obs), (rand, ep)
        a.check(); b.check()


def test_rollout_buffer_and_unaligned_outputs(torch_cuda):
    """step_into() writes rows into arbitrary (4-byte aligned) slices of a rollout buffer: odd sizes exercise the
    alignment phase of the LDS copy-out; rollout() == the same steps issued one by one."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    from deepcomp_amd.sharded import RolloutBuffer
    scn = scenarios.medium_map('mixed').with_ues(num_slow=2, num_fast=1)        # U = 3, B = 3: 13-float rows
    m, bs, ues = build_from_scenario(scn)
    T, E = 7, 5
    g = torch.Generator(device='cuda').manual_seed(5)
    acts = torch.randint(0, 4, (T, E, 3), generator=g, device='cuda', dtype=torch.uint8)

    ref = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=9, rng='philox')
    ref.reset()
    want_obs, want_rew = [], []
    for t in range(T):
        o, r, _, _ = ref.step(acts[t])
        want_obs.append(o.clone()); want_rew.append(r.clone())

    env = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=9, rng='philox')
    env.reset()
    buf = RolloutBuffer(env, T)
    assert (buf.obs[1].data_ptr() - buf.obs[0].data_ptr()) % 16 != 0           # 5*3*13*4 B = 780 B: misaligned slices
    it = iter(range(T))
    frag = buf.collect(lambda obs: acts[next(it)])
    env.check()
    assert torch.equal(frag['obs'], torch.stack(want_obs)) and torch.equal(frag['reward'], torch.stack(want_rew))

    env2 = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=9, rng='philox')
    env2.reset()
    o, r = env2.rollout(acts)
    assert torch.equal(o, want_obs[-1]) and torch.equal(r, want_rew[-1]) and env2.time == T


def test_reference_surface_single_env(torch_cuda):
    """The drop-in classes (E = 1) return the reference's Python structures and match a golden trajectory."""
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import make_env_config
    from deepcomp_amd.env import CentralRelNormEnv, MultiAgentMobileEnv
    g = np.load(os.path.join(GOLDEN, 'traj_custom4x4_multi_s42.npz'))
    env = MultiAgentMobileEnv(make_env_config(scenarios.custom_map('mixed').with_ues(num_slow=4), seed=42))
    obs = env.reset()
    assert sorted(obs.keys()) == ['1', '2', '3', '4'] and sorted(obs['1'].keys()) == ['connected', 'dr', 'ues_at_bs', 'util_at_bs', 'utility']
    for t in range(30):
        obs, rew, done, info = env.step({ue.id: int(g['actions'][t][i]) for i, ue in enumerate(env.ue_list)})
        assert done == {'1': None, '2': None, '3': None, '4': None, '__all__': None}
        np.testing.assert_allclose([rew[str(i + 1)] for i in range(4)], g['step_reward'][t], atol=ATOL_UTIL)
        assert [obs[str(i + 1)]['connected'] for i in range(4)] == g['step_obs_connected'][t].astype(int).tolist()
        assert info['1']['time'] == t + 1
        assert info['1']['scalar_metrics']['sum_utility'] == pytest.approx(float(g['step_sum_utility'][t]), abs=4e-4)
    assert env.ue_list[0].pos.x == g['step_pos'][29][0][0]
    g = np.load(os.path.join(GOLDEN, 'traj_medium3x3_central_s42.npz'))
    env = CentralRelNormEnv(make_env_config(scenarios.medium_map('mixed').with_ues(num_slow=3), seed=42))
    obs = env.reset()
    assert len(obs['connected']) == 9 and len(obs['dr']) == 9 and len(obs['utility']) == 3
    for t in range(30):
        obs, rew, done, info = env.step([int(x) for x in g['actions'][t]])
        assert done is None and isinstance(rew, float)
        assert rew == pytest.approx(float(g['step_reward'][t][0]), abs=ATOL_OBS)
        assert obs['connected'] == g['step_obs_connected'][t].astype(int).reshape(-1).tolist()
    with pytest.raises(AssertionError):
        env.step([4, 0, 0])                          # central.py:61


@pytest.mark.parametrize('dyn', [False, True])
def test_checkpoint_resume_is_bit_identical(torch_cuda, dyn):
    """state_dict() / load_state_dict(): a second env batch built from the same config continues a checkpoint taken mid-episode
    bit-identically (state, observations, rewards), across the next reset too -- with UE arrival / departure as well."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    E, L = 257, 40
    scn = scenarios.large_map('mixed').with_ues(num_static=1, num_slow=4, num_fast=2)
    m, bs, ues = build_from_scenario(scn)
    kw = dict(num_envs=E, seed=11, episode_length=L, rng='philox', rand_episodes=True, env_id_base=3,
              ue_arrival={3: 2, 9: -3, 15: 4, 22: -2, 31: 1} if dyn else None)
    a = BatchedMobileEnv(m, bs, ues, 'multi', **kw)
    rng = np.random.default_rng(5)
    acts = torch.from_numpy(rng.integers(0, a.B + 1, size=(70, E, a.U)).astype(np.uint8)).cuda()

    def run(env, t0, t1, record):
        for t in range(t0, t1):
            if t % L == 0:
                env.reset()
            env.step(acts[t])
            if record is not None:
                record.append((env.obs.cpu().numpy().copy(), env.reward.cpu().numpy().copy()))
        env.check()

    run(a, 0, 17, None)
    sd = a.state_dict()
    want = []
    run(a, 17, 70, want)                      # crosses the resets at t = 40
    b = BatchedMobileEnv(m, bs, ues, 'multi', **kw)
    b.load_state_dict(sd)
    assert b.time == 17 and b.num_ue == sd['counters'][2]
    got = []
    run(b, 17, 70, got)
    for i, ((o1, r1), (o2, r2)) in enumerate(zip(want, got)):
        assert np.array_equal(o1, o2) and np.array_equal(r1, r2), f'step {17 + i} differs after resume'
    sa, sb = a.state_host(), b.state_host()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    with pytest.raises(ValueError):
        BatchedMobileEnv(m, bs, ues, 'multi', **dict(kw, seed=12)).load_state_dict(sd)


def test_checkpoint_between_seed_and_reset(torch_cuda):
    """seed(s) on a Philox env is STAGED until reset(); a checkpoint taken in between must continue the running episode on
    the OLD key and switch to the new one at the next reset -- the checkpoint carries the device's key and the pending seed apart."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    E, L = 65, 12
    m, bs, ues = build_from_scenario(scenarios.medium_map('mixed').with_ues(num_slow=3, num_fast=2))
    kw = dict(num_envs=E, seed=11, episode_length=L, rng='philox', rand_episodes=True)
    acts = torch.randint(0, len(bs) + 1, (40, E, 5), device='cuda', dtype=torch.uint8)

    def run(env, t0, t1, rec):
        for t in range(t0, t1):
            if t % L == 0:
                env.reset()
            env.step(acts[t])
            rec.append(env.obs.clone())
    a = BatchedMobileEnv(m, bs, ues, 'multi', **kw)
    run(a, 0, 5, [])
    a.seed(4242)
    sd = a.state_dict()
    assert sd['config']['seed'] == 11 and sd['pending_seed'] == 4242
    want, got = [], []
    run(a, 5, 30, want)
    b = BatchedMobileEnv(m, bs, ues, 'multi', **kw)              # configured like the checkpointed env was
    b.load_state_dict(sd)
    run(b, 5, 30, got)
    for i, (x, y) in enumerate(zip(want, got)):
        assert torch.equal(x, y), f'step {5 + i}'
    # the second episode runs on the new key: an env constructed with it agrees from its first reset on
    c = BatchedMobileEnv(m, bs, ues, 'multi', **dict(kw, seed=4242))
    c.reset()
    c.step(acts[12])
    assert torch.equal(c.obs, want[12 - 5])
    # a seed staged in the RESTORING env is not part of the checkpointed run
    d = BatchedMobileEnv(m, bs, ues, 'multi', **kw)
    d.seed(5)
    sd0 = BatchedMobileEnv(m, bs, ues, 'multi', **kw)
    run(sd0, 0, 3, [])
    d.load_state_dict(sd0.state_dict())
    x, y = [], []
    run(sd0, 3, 20, x); run(d, 3, 20, y)
    assert all(torch.equal(p, q) for p, q in zip(x, y))
    # velocities are part of the fingerprint now
    m2, bs2, ues2 = build_from_scenario(scenarios.medium_map('mixed').with_ues(num_slow=2, num_fast=3))
    with pytest.raises(ValueError):
        BatchedMobileEnv(m2, bs2, ues2, 'multi', **kw).load_state_dict(sd)


def test_checkpoint_across_the_packing_boundary(torch_cuda, monkeypatch):
    """VERDICT r3 housekeeping: a checkpoint of a tightly packed batch (U lanes per env, scan-order sums) restored into the same
    batch with padded lane groups (butterfly-order sums) -- DCOMP_TIGHT overrides dcomp_create's choice; otherwise the same
    configuration always packs the same way, and load_state_dict() refuses another shape.  The state tensors do not depend on the
    packing: positions, movement words and masks continue BIT-identically, floats within the summation-order difference
    (<= 2e-6 relative)."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    E, U, B = 8192, 10, 5
    m, bs, ues = build_from_scenario(scenarios.grid_map(B, 'mixed').with_ues(num_slow=6, num_fast=4))
    kw = dict(num_envs=E, seed=5, episode_length=30, rng='philox', rand_episodes=True)
    acts = torch.randint(0, B + 1, (40, E, U), device='cuda', dtype=torch.uint8)
    monkeypatch.setenv('DCOMP_TIGHT', '1')
    a = BatchedMobileEnv(m, bs, ues, 'central', **kw)
    assert a.lanes_per_env == U
    a.reset()
    for t in range(12):
        a.step(acts[t])
    sd = a.state_dict()
    monkeypatch.setenv('DCOMP_TIGHT', '0')
    b = BatchedMobileEnv(m, bs, ues, 'central', **kw)
    assert b.lanes_per_env == 16
    b.load_state_dict(sd)
    for t in range(12, 40):
        if t == 30:
            a.reset(); b.reset()
        a.step(acts[t]); b.step(acts[t])
        assert torch.equal(a.pos, b.pos) and torch.equal(a.mv, b.mv) and torch.equal(a.conn, b.conn), t
        torch.testing.assert_close(a.ewma, b.ewma, rtol=1e-5, atol=0)
        torch.testing.assert_close(a.obs, b.obs, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(a.reward, b.reward, rtol=0, atol=2e-5)
    a.check(); b.check()


def test_wide_kernel_with_capped_occupancy_is_the_same_kernel(torch_cuda, monkeypatch):
    """Batches whose observation rows exceed 1 GB per step launch the wide kernel with 24 000 B of unused dynamic LDS per workgroup (two
    resident workgroups per CU instead of four: dcomp_create).  Same code, same results: forced on for a small batch and compared bit for
    bit with the default launch."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    E, U, B = 37, 128, 32
    m, bs, ues = build_from_scenario(scenarios.grid_map(B, 'mixed').with_ues(num_slow=100, num_fast=28))
    acts = torch.randint(0, B + 1, (12, E, U), device='cuda', dtype=torch.uint8)
    outs = []
    for pad in ('0', '24000'):
        monkeypatch.setenv('DCOMP_WIDE_PAD_LDS', pad)
        env = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=3, rng='philox')
        assert env.step_kernel_name == 'step_kernel_wide<32, 128, 2>'
        env.reset()
        for t in range(12):
            env.step(acts[t])
        env.check()
        outs.append((env.obs.clone(), env.reward.clone(), env.pos.clone(), env.mv.clone(), env.conn.clone(), env.ewma.clone()))
    for x, y in zip(*outs):
        assert torch.equal(x, y)


def test_episode_horizon_guard(torch_cuda):
    """The draw cursor and conn_since are 16-bit: a step beyond 65536 is refused (NotImplementedError), reset() clears it."""
    import ctypes as C
    torch = torch_cuda
    from deepcomp_amd import _lib, scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    m, bs, ues = build_from_scenario(scenarios.small_map('mixed').with_ues(num_slow=2))
    env = BatchedMobileEnv(m, bs, ues, 'central', num_envs=4, seed=1, rng='philox', rand_episodes=True, episode_length=10 ** 6)
    env.reset()
    a = torch.zeros((4, 2), dtype=torch.uint8, device='cuda')
    env.step(a)
    c = (C.c_int64 * 5)()
    _lib.check(env._L.dcomp_get_counters(env._h, c))
    c[0] = 65535
    _lib.check(env._L.dcomp_set_counters(env._h, c))
    env.step(a)                                        # step 65536 of the episode: the last one allowed
    with pytest.raises(NotImplementedError):
        env.step(a)
    env.reset()
    env.step(a)
    env.check()


def test_policy_map_helpers(torch_cuda):
    """get_max_num_ue / get_num_diff_ues (base.py:191-225): env_setup.py:295 sizes the --separate-agent-nns policy map with them."""
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import make_env_config
    from deepcomp_amd.env import CentralRelNormEnv, MultiAgentMobileEnv
    scn = scenarios.medium_map('mixed').with_ues(num_slow=3)
    env = MultiAgentMobileEnv(make_env_config(scn, episode_length=12, ue_arrival={'2': 1, '5': -1, '8': 2}))
    assert env.get_max_num_ue() == 5 and env.max_ues == 5          # 3 -> 4 -> 3 -> 5
    assert env.get_num_diff_ues() == 6                             # 3 + 1 + 2: a departure does not free an id
    env2 = CentralRelNormEnv(make_env_config(scn, episode_length=11, new_ue_interval=4))
    assert env2.get_max_num_ue() == 3 + int(10 / 4) == env2.get_num_diff_ues()
    env3 = MultiAgentMobileEnv(make_env_config(scn))
    assert env3.get_max_num_ue() == env3.get_num_diff_ues() == 3
    with pytest.raises(NotImplementedError):
        env3.render()


def test_step_kernel_name_is_what_the_profiles_say(torch_cuda):
    """dcomp_step_kernel_name: the instantiation dcomp_step launches, spelled as rocprofv3 prints it -- bench.py only reports a
    tracked --pmc traffic figure for the kernel the library really dispatches to (profiles/traffic.json entries name theirs)."""
    import json
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    db = json.load(open(os.path.join(os.path.dirname(GOLDEN), '..', 'profiles', 'traffic.json')))
    for key, (E, U, B, kind) in {'65536x32x10_multi_mixed': (65536, 32, 10, 'multi'), '32768x32x10_multi_mixed': (32768, 32, 10, 'multi'),
                                 '4096x128x32_multi_mixed': (4096, 128, 32, 'multi'), '65536x10x5_central_mixed': (65536, 10, 5, 'central')}.items():
        m, bs, ues = build_from_scenario(scenarios.grid_map(B, 'mixed').with_ues(num_slow=U))
        env = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=1, rng='philox')
        assert env.step_kernel_name == db[key]['kernel'], key
        del env
    m, bs, ues = build_from_scenario(scenarios.grid_map(5, 'resource-fair').with_ues(num_slow=10))
    assert BatchedMobileEnv(m, bs, ues, 'multi', num_envs=64, seed=1, rng='philox').step_kernel_name == 'step_kernel<5, 16, 1>'
