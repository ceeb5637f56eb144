"""More than 32 base stations (round 5).  The reference has no station limit (station.py:16-30, base.py:79-84); the native env had 32 (one
32-bit connection mask per UE, one kernel instantiation per station count).  33 ... 64 stations now take the GENERIC kernel of
deepcomp_amd/csrc/dcomp_big.h (run-time B, per-UE rows in LDS, the connection set in state.conn + state.conn_hi).  Held to:

* the reference itself -- two reference-run trajectories with 270 / 300 UEs in one env (tests/golden/traj_crowd*) and three with 36 / 40 / 64 stations (tests/golden/traj_dense*: static UEs parked at
  stations >= 32, scripted connects / disconnects, a max-cap rate tie) go through test_parity_gpu.py::test_golden_trajectory like every
  other traj_* fixture, and through tests/test_oracle_golden.py on the CPU;
* the oracle, on Philox batches at B = 33 ... 64 across lane widths, env kinds, reward aggregations and sharing models (here);
* the specialised kernels, with DCOMP_FORCE_BIG=1 at B <= 32: masks / positions identical, floats to the last bits (here).
"""
import numpy as np
import pytest

from tests import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _scenario(U, B, sharing, pitch=45):
    from deepcomp_amd import scenarios
    n_static = max(1, U // 8)
    scn = scenarios.grid_map(B, 'mixed' if sharing == 'mixed' else sharing, pitch=pitch, border=25).with_ues(
        num_static=n_static, num_slow=U - n_static - U // 4, num_fast=U // 4)
    if sharing == 'max-cap':                                  # a few other models among the max-cap stations, one of them beyond 31
        for b, m in ((1, 'resource-fair'), (B - 1, 'rate-fair'), (B - 2, 'proportional-fair')):
            scn.bs_sharing[b] = m
    return scn


def _oracle_batch(scn, kind, reward, E, seed):
    from oracle import oracle as orc
    envs = []
    for e in range(E):
        o = orc.OracleEnv(int(scn.width), int(scn.height), scn.bs_pos, scn.bs_sharing, [s['velocity'] for s in scn.ue_specs],
                          kind=orc.MULTI if kind == 'multi' else orc.CENTRAL, reward_agg={'avg': 0, 'sum': 1, 'min': 2}[reward])
        o.set_philox(seed, e)
        envs.append(o)
    return orc.OracleBatch(envs)


def _near_actions(rng, core, B, frac=0.6):
    """Actions biased towards stations that are in range (so that UEs hold several connections, high stations included)."""
    E, U = core.E, core.U
    pos = core.state_host()['pos']                                             # [E, U, 2]
    bs = np.stack([core._bs_x, core._bs_y], axis=1)                            # [B, 2]
    d = np.linalg.norm(pos[:, :, None, :] - bs[None, None, :, :], axis=-1)     # [E, U, B]
    near = d < 68.0
    a = rng.integers(0, B + 1, size=(E, U))
    pick = np.where(near.any(-1), (near * rng.random((E, U, B))).argmax(-1) + 1, a)
    a = np.where(rng.random((E, U)) < frac, pick, a)
    a[rng.random((E, U)) < 0.15] = 0
    return a.astype(np.uint8)


SHAPES = [('multi', 32, 64, 96, 'avg', 'mixed'), ('central', 10, 40, 128, 'avg', 'mixed'), ('multi', 3, 33, 200, 'min', 'mixed'),
          ('multi', 12, 48, 64, 'sum', 'mixed'), ('multi', 70, 36, 12, 'avg', 'mixed'), ('central', 130, 40, 6, 'min', 'mixed'),
          ('multi', 128, 64, 8, 'avg', 'mixed'), ('multi', 200, 56, 3, 'sum', 'mixed'), ('central', 256, 33, 2, 'sum', 'mixed'),
          ('multi', 20, 40, 48, 'avg', 'max-cap'), ('central', 9, 64, 64, 'sum', 'max-cap'), ('multi', 64, 64, 10, 'min', 'rate-fair'),
          ('multi', 16, 50, 64, 'avg', 'proportional-fair'), ('multi', 1, 64, 128, 'avg', 'mixed'), ('central', 5, 37, 300, 'avg', 'resource-fair')]


@pytest.mark.parametrize('shape', SHAPES)
def test_generic_kernel_against_the_oracle(torch_cuda, shape):
    """reset + 14 steps against the FP64 oracle: connection sets (64-bit), FSM state and FP64 positions bit-exact; rates / EWMA / obs.dr
    within 1e-5 relative, the utility-scaled entries and rewards at the bars of tests/parity.py -- the same assertions as for <= 32 stations."""
    torch = torch_cuda
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    kind, U, B, E, reward, sharing = shape
    scn = _scenario(U, B, sharing)
    m, bs, ues = build_from_scenario(scn)
    core = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=77, reward=reward, rng='philox')
    assert core.step_kernel_name.startswith('big_kernel<') and core.conn_hi is not None
    ob = _oracle_batch(scn, kind, reward, E, 77)
    rng = np.random.default_rng(B * 1000 + U)
    core.reset()
    parity.assert_step(core, ob, ob.reset(), None, None, None, kind, reward, msg='reset')
    high = 0
    for t in range(14):
        a = _near_actions(rng, core, B)
        core.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, o_conn, o_pos = ob.step(a)
        parity.assert_step(core, ob, o_obs, o_rew, o_conn, o_pos, kind, reward, msg=f'step {t}')
        high += int((o_conn >> np.uint64(32) != 0).sum())
    core.check()
    assert high > 0, 'no connection to a station beyond 31 was ever made: the scenario does not test the second mask word'
    assert np.array_equal(core.conn_hi.cpu().numpy().astype(np.uint32).reshape(E, U), (o_conn >> np.uint64(32)).astype(np.uint32))


@pytest.mark.parametrize('shape', [('multi', 300, 10, 6, 'avg', 'mixed'), ('central', 512, 8, 3, 'min', 'mixed'), ('multi', 1000, 5, 2, 'sum', 'mixed'),
                                   ('multi', 400, 40, 2, 'avg', 'max-cap'), ('central', 257, 32, 4, 'sum', 'rate-fair'), ('multi', 1024, 12, 1, 'min', 'mixed')])
def test_more_than_256_ues_per_env_against_the_oracle(torch_cuda, shape):
    """The other size limit the reference does not have (base.py:79-84): 257 ... 1 024 UEs in ONE env -- a workgroup of up to 1 024 lanes of the
    generic kernel.  Same assertions as everywhere: masks / positions bit-exact, floats at the bars of tests/parity.py."""
    torch = torch_cuda
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    kind, U, B, E, reward, sharing = shape
    scn = _scenario(U, B, sharing)
    m, bs, ues = build_from_scenario(scn)
    core = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=31, reward=reward, rng='philox')
    assert core.step_kernel_name.startswith('big_kernel<') and core.lanes_per_env in (512, 1024) and core.conn_hi is not None
    ob = _oracle_batch(scn, kind, reward, E, 31)
    rng = np.random.default_rng(U)
    core.reset()
    parity.assert_step(core, ob, ob.reset(), None, None, None, kind, reward, msg='reset')
    for t in range(8):
        a = _near_actions(rng, core, B)
        core.step(torch.from_numpy(a).cuda())
        parity.assert_step(core, ob, *ob.step(a), kind, reward, msg=f'step {t}')
    core.check()
    if shape is not None and U == 1024:
        # round 6: nothing per (UE, station) lives in LDS any more -- 1 024 UE slots x 64 stations (refused until round 5: its rows did not fit
        # 160 KB) is one workgroup like any other
        scn2 = _scenario(1024, 64, 'mixed')
        full = BatchedMobileEnv(*build_from_scenario(scn2), 'multi', num_envs=1, seed=9, rng='philox')
        ob2 = _oracle_batch(scn2, 'multi', 'avg', 1, 9)
        full.reset()
        parity.assert_step(full, ob2, ob2.reset(), None, None, None, 'multi', 'avg', msg='1024 x 64 reset')
        for t in range(3):
            a = _near_actions(rng, full, 64)
            full.step(torch.from_numpy(a).cuda())
            parity.assert_step(full, ob2, *ob2.step(a), 'multi', 'avg', msg=f'1024 x 64 step {t}')
        full.check()
        with pytest.raises(ValueError):
            BatchedMobileEnv(*build_from_scenario(_scenario(1025, 4, 'mixed')), 'multi', num_envs=1, rng='philox')


@pytest.mark.parametrize('kind,U,B,E,reward,sharing', [('multi', 32, 10, 64, 'avg', 'mixed'), ('central', 10, 5, 100, 'avg', 'mixed'),
                                                       ('multi', 128, 32, 4, 'min', 'mixed'), ('multi', 20, 7, 40, 'sum', 'max-cap'),
                                                       ('central', 70, 12, 9, 'min', 'rate-fair')])
def test_generic_kernel_equals_the_specialised_kernels(torch_cuda, kind, U, B, E, reward, sharing, monkeypatch):
    """DCOMP_FORCE_BIG=1 sends station counts the specialised kernels serve through the generic one: positions, movement words and
    connection sets identical after 25 steps, every float within float32 rounding of the other path (the sums run in another order)."""
    torch = torch_cuda
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    scn = _scenario(U, B, sharing, pitch=60)
    m, bs, ues = build_from_scenario(scn)
    mk = lambda: BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=5, reward=reward, rng='philox')       # noqa: E731
    ref = mk()
    monkeypatch.setenv('DCOMP_FORCE_BIG', '1')
    big = mk()
    monkeypatch.delenv('DCOMP_FORCE_BIG')
    assert big.step_kernel_name.startswith('big_kernel<') and not ref.step_kernel_name.startswith('big_kernel<')
    rng = np.random.default_rng(3)
    ref.reset(); big.reset()
    assert torch.equal(ref.pos, big.pos) and torch.equal(ref.mv, big.mv)
    torch.testing.assert_close(big.obs, ref.obs, rtol=2e-6, atol=2e-6)
    for t in range(25):
        a = torch.from_numpy(_near_actions(rng, ref, B)).cuda()
        ref.step(a); big.step(a)
        assert torch.equal(ref.pos, big.pos) and torch.equal(ref.mv, big.mv) and torch.equal(ref.conn, big.conn), f'step {t}'
        torch.testing.assert_close(big.obs, ref.obs, rtol=3e-6, atol=3e-6, msg=lambda s_: f'step {t} obs: {s_}')
        torch.testing.assert_close(big.reward, ref.reward, rtol=0, atol=2e-5 * (U if reward == 'sum' else 1))
        torch.testing.assert_close(big.ewma, ref.ewma, rtol=3e-6, atol=1e-30)
        torch.testing.assert_close(big.ue_dr, ref.ue_dr, rtol=3e-6, atol=1e-30)
        torch.testing.assert_close(big.sum_utility, ref.sum_utility, rtol=0, atol=2e-5 * U)
    assert int(big.conn_hi.abs().sum()) == 0
    ref.check(); big.check()


@pytest.mark.parametrize('kind,U,B,E,sharing,reward,rng', [('multi', 32, 64, 24, 'mixed', 'avg', 'philox'), ('central', 10, 40, 50, 'mixed', 'sum', 'philox'),
                                                          ('multi', 12, 33, 9, 'max-cap', 'min', 'philox'), ('multi', 300, 12, 3, 'mixed', 'sum', 'philox'),
                                                          ('central', 600, 7, 2, 'proportional-fair', 'avg', 'philox'), ('multi', 70, 50, 3, 'rate-fair', 'avg', 'philox'),
                                                          ('multi', 6, 36, 5, 'mixed', 'avg', 'reference'), ('central', 5, 64, 4, 'resource-fair', 'min', 'reference')])
def test_generic_kernel_fused_rollout_equals_single_steps(torch_cuda, kind, U, B, E, sharing, reward, rng, monkeypatch):
    """big_kernel<..., ROLL>: T steps in one launch per stretch of an episode (UE state in registers in between, step t's outputs in slice t)
    against the same steps issued one by one with reset() at the horizon: every step's outputs, the last-step-only form, the compact record, the
    closed policy loop and the final state, bit for bit; DCOMP_NO_FUSED_BIG=1 (one launch per step inside the same call) gives the same again."""
    torch = torch_cuda
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    from deepcomp_amd.fragment import FragmentCodec
    L, T = 9, 22
    scn = _scenario(U, B, sharing)
    m, bs, ues = build_from_scenario(scn)
    mk = lambda: BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=21, rng=rng, rand_episodes=(rng == 'philox'), episode_length=L, reward=reward)
    ref, env, last, comp = mk(), mk(), mk(), mk()
    assert env.step_kernel_name.startswith('big_kernel<') and env.fused_rollout and env.rollout_is_fused(T)
    nprng = np.random.default_rng(5)
    for e_ in (ref, env, last, comp):
        e_.reset()
    acts = torch.from_numpy(np.stack([_near_actions(nprng, ref, B) for _ in range(T)])).cuda()
    keys = ('obs', 'reward', 'sum_utility', 'ue_dr', 'ue_utility')
    want = {k: torch.empty((T,) + tuple(getattr(ref, k).shape), device='cuda') for k in keys}
    for t in range(T):
        if ref.time == L:
            ref.reset()
        ref.step(acts[t])
        for k in keys:
            want[k][t].copy_(getattr(ref, k))
    got = {k: torch.full_like(v, float('nan')) for k, v in want.items()}
    env.rollout(acts, out=got, horizon=L)
    for k in keys:
        assert torch.equal(got[k], want[k]), k
    state = ('pos', 'mv', 'conn', 'conn_hi', 'ewma')
    for k in state:
        assert torch.equal(getattr(env, k), getattr(ref, k)), k
    assert env.time == ref.time and env.episode == ref.episode
    last.rollout(acts, horizon=L)                                  # outputs of the last step only
    assert torch.equal(last.obs, want['obs'][-1]) and torch.equal(last.reward, want['reward'][-1]) and torch.equal(last.pos, ref.pos)
    if kind == 'multi':                                            # every step's compact record straight from the fused kernel
        codec = FragmentCodec(U, B)
        packed = {'obs_compact': torch.empty((T, E, codec.words), dtype=torch.int32, device='cuda'), 'reward': torch.empty_like(want['reward'])}
        comp.rollout(acts, out=packed, horizon=L)
        assert torch.equal(codec.unpack(packed['obs_compact']).view(torch.int32), want['obs'].view(torch.int32)) and torch.equal(packed['reward'], want['reward'])
    monkeypatch.setenv('DCOMP_NO_FUSED_BIG', '1')
    steps = mk()
    assert not steps.rollout_is_fused(T)
    steps.reset()
    got2 = {k: torch.full_like(v, float('nan')) for k, v in want.items()}
    steps.rollout(acts, out=got2, horizon=L)
    for k in keys:
        assert torch.equal(got2[k], want[k]), k
    monkeypatch.delenv('DCOMP_NO_FUSED_BIG')
    # the closed loop: fused (decisions read back from next_action inside the launch) against heuristic_actions + step
    a, b = mk(), mk()
    a.reset()
    assert b.set_policy('dynamic', 0.4)
    b.reset()
    w_obs, w_rew = [], []
    for t in range(T):
        if a.time == L:
            a.reset()
        a.step(a.heuristic_actions('dynamic', 0.4))
        w_obs.append(a.obs.clone()); w_rew.append(a.reward.clone())
    out = {'obs': torch.full((T,) + tuple(b.obs.shape), float('nan'), device='cuda'), 'reward': torch.empty((T,) + tuple(b.reward.shape), device='cuda')}
    b.rollout_policy(T, out=out, horizon=L)
    assert torch.equal(out['obs'], torch.stack(w_obs)) and torch.equal(out['reward'], torch.stack(w_rew))
    for k in state:
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert torch.equal(b.next_action, a.heuristic_actions('dynamic', 0.4))
    for e_ in (ref, env, last, comp, steps, a, b):
        e_.check()


@pytest.mark.parametrize('U,B,E,arrival', [(32, 64, 9, None), (13, 33, 21, None), (10, 47, 7, None), (70, 40, 3, None), (300, 36, 2, None), (7, 50, 11, {1: 3, 3: -2, 5: 4, 8: -5})])
def test_generic_kernel_rows_as_16_byte_stores_equal_the_four_block_form(torch_cuda, U, B, E, arrival, monkeypatch):
    """BigParams::row_x4 (multi-agent rows of more than 32 stations transposed through one LDS row per wavefront and written as one 16-byte
    store per lane -- what dcomp_create picks once a step's rows exceed the Infinity Cache) against the four-block form on twin envs: reset,
    steps, a fused rollout fragment, the in-step policy and UE arrival / departure (zero rows of unlisted slots), every output bit for bit.
    Station counts that are not multiples of 4 and UE counts that leave rows at 4-byte alignment included."""
    torch = torch_cuda
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    scn = _scenario(U, B, 'mixed')
    m, bs, ues = build_from_scenario(scn)
    kw = dict(num_envs=E, seed=17, rng='philox', rand_episodes=True, episode_length=12)
    if arrival:
        kw.update(ue_arrival=arrival, max_ues=U + 6)
    envs = []
    for flag in ('0', '1'):
        monkeypatch.setenv('DCOMP_BIG_ROW_X4', flag)
        envs.append(BatchedMobileEnv(m, bs, ues, 'multi', **kw))
    a, b = envs
    assert a.step_kernel_name.startswith('big_kernel<')
    keys = ('obs', 'reward', 'sum_utility', 'ue_dr', 'ue_utility')
    same = lambda tag: [None for k in keys if not torch.equal(getattr(a, k).view(torch.int32), getattr(b, k).view(torch.int32)) and pytest.fail(f'{tag}: {k}')]
    a.reset(); b.reset()
    same('reset')
    rng = np.random.default_rng(3)
    for t in range(10):
        act = torch.from_numpy(rng.integers(0, B + 1, size=(E, a.U)).astype(np.uint8)).cuda()
        a.step(act); b.step(act)
        same(f'step {t}')
    a.reset(); b.reset()
    assert a.set_policy('dynamic', 0.3) and b.set_policy('dynamic', 0.3)
    a.reset(); b.reset()
    T = 9
    outs = [{'obs': torch.full((T,) + tuple(e_.obs.shape), float('nan'), device='cuda'), 'reward': torch.empty((T,) + tuple(e_.reward.shape), device='cuda')} for e_ in envs]
    a.rollout_policy(T, out=outs[0]); b.rollout_policy(T, out=outs[1])
    assert torch.equal(outs[0]['obs'].view(torch.int32), outs[1]['obs'].view(torch.int32)) and torch.equal(outs[0]['reward'], outs[1]['reward'])
    assert torch.equal(a.next_action, b.next_action) and torch.equal(a.pos, b.pos) and torch.equal(a.conn_hi, b.conn_hi)
    a.check(); b.check()


def test_many_stations_what_works_and_what_says_no(torch_cuda):
    """48 stations: rollout() (fused: one launch per stretch of an episode) == step(), checkpoints carry conn_hi, bad actions are flagged; the features that
    live in the specialised kernels only say so (UE arrival / departure, in-step policy, compact record, the fragment codec)."""
    torch = torch_cuda
    from deepcomp_amd import fragment
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    U, B, E = 12, 48, 40
    scn = _scenario(U, B, 'mixed')
    m, bs, ues = build_from_scenario(scn)
    kw = dict(num_envs=E, seed=9, rng='philox', rand_episodes=True, episode_length=20)
    a, b = BatchedMobileEnv(m, bs, ues, 'multi', **kw), BatchedMobileEnv(m, bs, ues, 'multi', **kw)
    rng = np.random.default_rng(1)
    a.reset(); b.reset()
    T = 30                                                        # across the horizon of 20
    acts = torch.from_numpy(np.stack([_near_actions(rng, a, B) for _ in range(T)])).cuda()
    assert a.rollout_is_fused(T) and a.fused_rollout            # (round 6: the generic kernel's fused rollout, one launch per stretch of an episode)
    want = {'obs': torch.empty((T,) + tuple(a.obs.shape), device='cuda'), 'reward': torch.empty((T,) + tuple(a.reward.shape), device='cuda')}
    got = {k: torch.empty_like(v) for k, v in want.items()}
    for t in range(T):
        if a.time == 20:
            a.reset()
        a.step_into(acts[t], want['obs'][t], want['reward'][t])
    b.rollout(acts, out=got, horizon=20)
    assert torch.equal(got['obs'], want['obs']) and torch.equal(got['reward'], want['reward'])
    assert torch.equal(a.pos, b.pos) and torch.equal(a.conn, b.conn) and torch.equal(a.conn_hi, b.conn_hi) and int(a.conn_hi.abs().sum()) > 0
    # checkpoint / resume with the second mask word
    sd = a.state_dict()
    c = BatchedMobileEnv(m, bs, ues, 'multi', **kw)
    c.reset()
    c.load_state_dict(sd)
    for t in range(5):
        a.step(acts[t]); c.step(acts[t])
    assert torch.equal(a.conn_hi, c.conn_hi) and torch.equal(a.obs, c.obs) and torch.equal(a.pos, c.pos)
    # action B + 1 is outside the space (base.py:238)
    bad = acts[0].clone(); bad[3, 2] = B + 1
    a.step(bad)
    with pytest.raises(AssertionError):
        a.check()
    # (round 6: the generic kernel carries the in-step policy too -- tests/test_adapters_gpu.py::test_in_step_policy_equals_the_policy_kernel)
    assert a.set_policy('3gpp') is True and a.step_kernel_name.endswith('false, true, false>')
    a.set_policy(None)
    assert fragment.fragment_words(U, B) == U * (B + 3) + 2 * B                    # (round 6: the compact record with two set words per UE)
    assert BatchedMobileEnv(m, bs, ues, 'multi', ue_arrival={3: 1}, **kw).step_kernel_name.startswith('big_kernel<')      # (round 6: UE arrival / departure too)
    with pytest.raises(ValueError):
        BatchedMobileEnv(*build_from_scenario(_scenario(4, 65, 'mixed')), 'multi', num_envs=2, rng='philox')


@pytest.mark.parametrize('kind,U,B,E', [('multi', 12, 48, 64), ('central', 9, 64, 40), ('multi', 70, 33, 6), ('multi', 32, 64, 16)])
def test_heuristic_policies_with_many_stations(torch_cuda, kind, U, B, E):
    """The reference's heuristic baselines (agent/heuristics.py:13-187) on observations of 33 ... 64 stations: dcomp_heuristic_actions
    (64-bit sets, two cluster-mask words per station) against the tensor-expression form of the rules, on live and on tie-ridden
    synthetic observations; a heuristic-driven closed loop (`agent.act(env)`) stays bit-exact with the oracle fed the same actions."""
    torch = torch_cuda
    from deepcomp_amd import agents
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    scn = _scenario(U, B, 'mixed')
    m, bs, ues = build_from_scenario(scn)
    env = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=4, rng='philox')
    ob = _oracle_batch(scn, kind, 'avg', E, 4)
    env.reset(); ob.reset()

    def views(obs):
        if kind == 'multi':
            return {'connected': obs[..., :B], 'dr': obs[..., B:2 * B]}
        return {'connected': obs[:, :U * B].reshape(E, U, B), 'dr': obs[:, U * B:2 * U * B].reshape(E, U, B)}
    ags = [agents.Heuristic3GPP(), agents.FullCoMP(), agents.DynamicSelection(0.3), agents.DynamicSelection(1.0),
           agents.StaticClustering(5, bs, seed=5, device='cuda')]
    high = 0
    for t in range(10):
        for ag in ags:
            assert torch.equal(ag.act(env), ag(views(env.obs))), (type(ag).__name__, t)
        a = ags[t % len(ags)].act(env)
        env.step(a)
        o_obs, o_rew, o_conn, o_pos = ob.step(a.cpu().numpy())
        st = env.state_host()
        assert np.array_equal(st['conn'], o_conn) and np.array_equal(st['pos'], o_pos), f'step {t}'
        high += int((o_conn >> np.uint64(32) != 0).sum())
    assert high > 0 and ags[-1]._bits.shape == (B, 2)
    env.check()
    g = torch.Generator(device='cuda').manual_seed(1)
    syn = torch.zeros_like(env.obs)
    v = views(syn)
    v['dr'].copy_(torch.randint(0, 4, v['dr'].shape, generator=g, device='cuda') / 3.0)           # quantised: ties everywhere
    v['connected'].copy_((torch.rand(v['connected'].shape, generator=g, device='cuda') < 0.3).float())
    v['connected'][0] = 1.0
    v['connected'][-1] = 0.0
    for ag, args in zip(ags, (('3gpp',), ('fullcomp',), ('dynamic', 0.3), ('dynamic', 1.0), ('cluster', 0.0, ags[-1]._bits))):
        assert torch.equal(env.heuristic_actions(*args, obs=syn), ag(views(syn))), type(ag).__name__


def test_many_stations_at_scale_against_the_oracle(torch_cuda):
    """4 096 x 32 UE x 64 stations (8.4 M pairs per step), 5 steps, every UE against the oracle."""
    torch = torch_cuda
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    E, U, B = 4096, 32, 64
    scn = _scenario(U, B, 'mixed')
    m, bs, ues = build_from_scenario(scn)
    core = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=42, rng='philox')
    ob = _oracle_batch(scn, 'multi', 'avg', E, 42)
    rng = np.random.default_rng(8)
    core.reset()
    parity.assert_step(core, ob, ob.reset(), None, None, None, 'multi', msg='reset')
    for t in range(5):
        a = _near_actions(rng, core, B)
        core.step(torch.from_numpy(a).cuda())
        parity.assert_step(core, ob, *ob.step(a), 'multi', msg=f'step {t}')
    core.check()


DYN_SHAPES = [('multi', 6, 40, 120, 'avg', 'mixed'), ('central', 5, 64, 90, 'avg', 'mixed'), ('multi', 9, 48, 40, 'min', 'max-cap'),
              ('central', 12, 36, 33, 'sum', 'proportional-fair'), ('multi', 60, 40, 6, 'sum', 'mixed'), ('multi', 3, 33, 200, 'avg', 'rate-fair'),
              ('central', 130, 40, 3, 'min', 'max-cap')]


@pytest.mark.parametrize('shape', DYN_SHAPES)
def test_generic_kernel_with_ue_arrival_and_departure(torch_cuda, shape):
    """UE arrival / departure (base.py:433-443, 592-618) with more than 32 stations (round 6; VERDICT r5 item 2: f4 existed on the specialised
    kernels only).  Philox-keyed departures (which UE leaves differs per env) and border points, two episodes, against the oracle: UE ids,
    64-bit connection sets and FP64 positions exact at every step, observations / rewards at the bars of tests/parity.py.  Lane groups from
    8 to 256 slots per env (the slot shift inside a wavefront and through LDS), max-cap stations (the step-of-connection rows travel with
    their UEs), every reward aggregation."""
    torch = torch_cuda
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    from oracle import oracle as orc
    from tests.parity import ATOL_OBS, ATOL_UTIL
    kind, U0, B, E, reward, sharing = shape
    L = 40
    arrival = {2: 3, 5: -2, 9: 4, 14: -3, 20: 2, 21: 2, 30: -4, 33: 5, 37: -6}
    scn = _scenario(U0, B, sharing)
    m, bs, ues = build_from_scenario(scn)
    core = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=5, episode_length=L, reward=reward, rng='philox', rand_episodes=True, ue_arrival=arrival)
    assert core.step_kernel_name.startswith('big_kernel<') and core.step_kernel_name.endswith('true, false, false, false>') and core.conn_hi is not None
    M = core.U
    sched = orc.arrival_schedule(L, arrival)
    oenvs = []
    for e in range(E):
        o = orc.OracleEnv(int(scn.width), int(scn.height), scn.bs_pos, scn.bs_sharing, [s_['velocity'] for s_ in scn.ue_specs],
                          kind=orc.MULTI if kind == 'multi' else orc.CENTRAL, reward_agg={'avg': 0, 'sum': 1, 'min': 2}[reward], max_ues=M)
        o.set_philox(5, e)
        oenvs.append(o)
    ob = orc.OracleBatch(oenvs)
    rng = np.random.default_rng(3)
    high = 0
    for ep in range(2):
        for o in oenvs:
            o.set_episode(ep)
        core.reset()
        want = ob.reset()
        parity.assert_obs(core.obs.cpu().numpy(), want, kind, M, B, msg=f'episode {ep} reset')
        for t in range(L):
            a = _near_actions(rng, core, B)
            n_rem, n_add = sched[t]
            if n_rem or n_add:
                for o in oenvs:
                    o.set_event_counts(n_rem, n_add)
            core.step(torch.from_numpy(a).cuda())
            o_obs, o_rew, o_conn, o_pos = ob.step(a)
            st = core.state_host()
            assert core.num_ue == oenvs[0].num_ue()
            assert np.array_equal(st['uid'], np.stack([o.uids() for o in oenvs])), f'step {t}: UE ids differ'
            assert np.array_equal(st['conn'], o_conn), f'episode {ep} step {t}: connection sets'
            assert np.array_equal(st['pos'], o_pos), f'episode {ep} step {t}: positions'
            parity.assert_obs(core.obs.cpu().numpy(), o_obs, kind, M, B, msg=f'episode {ep} step {t}')
            tol = (ATOL_UTIL if kind == 'multi' else ATOL_OBS) * (M if reward == 'sum' else 1)
            np.testing.assert_allclose(core.reward.cpu().numpy(), o_rew, atol=tol, rtol=0)
            high += int((o_conn >> np.uint64(32) != 0).sum())
    core.check()
    assert high > 0, 'no connection at a station >= 32: the scenario no longer exercises the second mask word'


@pytest.mark.parametrize('name', ['dyn_custom_multi_updown_s42', 'dyn_medium_central_largeupdown_s42', 'dyn_large_multi_2eps_rand_s42',
                                  'reseeddyn_custom_central_2eps_fixed_s43'])
def test_generic_kernel_on_the_reference_run_ue_arrival_fixtures(torch_cuda, name, monkeypatch):
    """DCOMP_FORCE_BIG=1: the reference-run dyn_* / reseeddyn_* fixtures (<= 7 stations; tape mode: host-drawn departures and border points,
    MobileEnv.seed() on a live env) through the GENERIC kernel's event phase -- the same assertions as on the specialised dynamic kernel.
    (dyn_dense40_* / dyn_dense36_* take the generic kernel by themselves: 40 / 36 stations.)"""
    from tests import test_parity_gpu as tp
    monkeypatch.setenv('DCOMP_FORCE_BIG', '1')
    tp.test_golden_dynamic_ue_trajectory(torch_cuda, name, False)
    tp.test_golden_dynamic_ue_trajectory(torch_cuda, name, True)


@pytest.mark.parametrize('U,B,E,arrival', [(32, 64, 40, None), (12, 40, 64, None), (7, 33, 50, {2: 3, 5: -2, 9: 2}), (130, 48, 3, None), (300, 36, 2, None)])
def test_generic_kernel_writes_the_compact_record(torch_cuda, U, B, E, arrival):
    """dcomp_out.obs_compact on the generic kernel (round 6: with more than 32 stations the record carries TWO connection words per UE:
    U (B + 3) + 2B words): reset / step / rollout write the record themselves, `unpack` of it is BIT-identical to the rows a twin env writes --
    with UE arrival / departure (unlisted slots are zero records), envs wider than a wavefront, and through dcomp_pack_fragment of the twin's rows."""
    torch = torch_cuda
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    from deepcomp_amd.fragment import FragmentCodec
    scn = _scenario(U, B, 'mixed')
    m, bs, ues = build_from_scenario(scn)
    kw = dict(num_envs=E, seed=21, rng='philox', rand_episodes=True, episode_length=30, ue_arrival=arrival)
    rows_env = BatchedMobileEnv(m, bs, ues, 'multi', **kw)
    comp_env = BatchedMobileEnv(m, bs, ues, 'multi', **kw)
    M = rows_env.U
    codec = FragmentCodec(M, B)
    assert comp_env.compact_words == codec.words == M * (B + 3) + 2 * B and comp_env.step_kernel_name.startswith('big_kernel<')
    packed = torch.empty((E, codec.words), dtype=torch.int32, device='cuda')
    rows_env.reset()
    comp_env.reset_compact(packed)
    assert torch.equal(codec.unpack(packed).view(torch.int32), rows_env.obs.view(torch.int32)), 'reset'
    rng = np.random.default_rng(5)
    high = 0
    for t in range(14):
        a = torch.from_numpy(_near_actions(rng, rows_env, B)).cuda()
        rows_env.step(a)
        comp_env.step_compact(a, packed, comp_env.reward)
        got = codec.unpack(packed)
        assert torch.equal(got.view(torch.int32), rows_env.obs.view(torch.int32)), f'step {t}: unpack(compact record) != rows'
        assert torch.equal(comp_env.reward, rows_env.reward) and torch.equal(comp_env.conn_hi, rows_env.conn_hi) and torch.equal(comp_env.pos, rows_env.pos)
        assert torch.equal(codec.pack(rows_env.obs), packed), f'step {t}: the step\'s record != dcomp_pack_fragment of the rows'
        high += int((rows_env.conn_hi != 0).sum())
    codec.check(); rows_env.check(); comp_env.check()
    assert high > 0
    # a whole fragment through rollout(out={'obs_compact': ...})
    T = 6
    acts = torch.from_numpy(np.stack([_near_actions(rng, rows_env, B) for _ in range(T)])).cuda()
    frag = {'obs_compact': torch.empty((T, E, codec.words), dtype=torch.int32, device='cuda'), 'reward': torch.empty((T, E, M), device='cuda')}
    want = torch.empty((T, E, M, 4 * B + 1), device='cuda')
    if arrival is None:
        comp_env.rollout(acts, out=frag)
        for t in range(T):
            rows_env.step(acts[t])
            want[t] = rows_env.obs
        assert torch.equal(codec.unpack(frag['obs_compact']).view(torch.int32), want.view(torch.int32))
