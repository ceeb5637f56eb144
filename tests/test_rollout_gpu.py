"""GPU tests of the fused T-step rollout (dcomp_rollout / dcomp_rollout_ex, one launch for T steps) and of the LDS-staged
central observation stores.  Bar: BIT-IDENTICAL to the same steps issued one dcomp_step at a time (which the parity
suite ties to the reference fixtures and the oracle); one case also runs against the CPU oracle directly.

Replaces the per-step evaluation loop of deepcomp/util/simulation.py:512-541 and RLlib's reset at the horizon
(deepcomp/util/env_setup.py:281)."""
import numpy as np
import pytest

from tests import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _make(kind, U, B, E, reward='avg', sharing='mixed', seed=77, rand_episodes=True, L=100, rng='philox'):
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    scn = scenarios.grid_map(B, sharing).with_ues(num_static=U // 8, num_slow=U - U // 8 - U // 4, num_fast=U // 4)
    m, bs, ues = build_from_scenario(scn)
    return BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=seed, reward=reward, rng=rng, rand_episodes=rand_episodes,
                            episode_length=L, env_id_base=3)


def _state(env):
    return {k: getattr(env, k).clone() for k in ('pos', 'mv', 'conn', 'ewma')}


def _same_state(a, b):
    import torch
    return all(torch.equal(a[k], b[k]) for k in a)


SHAPES = [('multi', 32, 10, 200, 'avg', 'mixed'), ('central', 10, 5, 333, 'avg', 'mixed'), ('multi', 7, 4, 129, 'sum', 'mixed'),
          ('multi', 20, 6, 64, 'min', 'max-cap'), ('central', 16, 9, 50, 'sum', 'rate-fair'), ('multi', 3, 3, 700, 'avg', 'resource-fair'),
          ('central', 32, 10, 65, 'min', 'mixed'), ('multi', 100, 12, 9, 'avg', 'mixed'), ('central', 12, 16, 40, 'avg', 'proportional-fair'),
          ('central', 64, 15, 11, 'avg', 'mixed'), ('central', 5, 32, 77, 'avg', 'mixed'), ('multi', 128, 32, 3, 'avg', 'mixed'),
          ('central', 130, 6, 5, 'avg', 'mixed'), ('central', 40, 20, 7, 'sum', 'max-cap')]


@pytest.mark.parametrize('shape', SHAPES)
def test_fused_rollout_equals_single_steps(torch_cuda, shape):
    """rollout(T) -- last-step outputs and [T, ...] fragments -- against T step() calls: every tensor bit-identical."""
    torch = torch_cuda
    kind, U, B, E, reward, sharing = shape
    T = 23
    g = torch.Generator(device='cuda').manual_seed(11)
    acts = torch.randint(0, B + 1, (T, E, U), generator=g, device='cuda', dtype=torch.uint8)
    acts[torch.rand((T, E, U), generator=g, device='cuda') < 0.4] = 0

    ref = _make(kind, U, B, E, reward, sharing)
    ref.reset()
    want = {k: [] for k in ('obs', 'reward', 'sum_utility', 'ue_dr', 'ue_utility')}
    for t in range(T):
        ref.step(acts[t])
        for k in want:
            want[k].append(getattr(ref, k).clone())
    ref.check()
    want = {k: torch.stack(v) for k, v in want.items()}

    a = _make(kind, U, B, E, reward, sharing)
    a.reset()
    o, r = a.rollout(acts)                                   # outputs of the last step only
    a.check()
    assert a.time == T
    assert torch.equal(o, want['obs'][-1]) and torch.equal(r, want['reward'][-1])
    assert torch.equal(a.sum_utility, want['sum_utility'][-1]) and torch.equal(a.ue_dr, want['ue_dr'][-1])
    assert _same_state(_state(a), _state(ref))

    b = _make(kind, U, B, E, reward, sharing)
    b.reset()
    out = {k: torch.full_like(v, float('nan')) for k, v in want.items()}
    b.rollout(acts[:10], out={k: v[:10] for k, v in out.items()})          # two fragments: the state carries over
    b.rollout(acts[10:], out={k: v[10:] for k, v in out.items()})
    b.check()
    for k in want:
        assert torch.equal(out[k], want[k]), k
    assert _same_state(_state(b), _state(ref))


@pytest.mark.parametrize('kind,U,B,E,rand', [('multi', 32, 10, 96, True), ('central', 10, 5, 200, True), ('multi', 9, 7, 40, False),
                                             ('multi', 128, 32, 3, True)])
def test_rollout_resets_at_the_horizon(torch_cuda, kind, U, B, E, rand):
    """horizon=L inside rollout() == `if time == L: reset()` before every step (RLlib's horizon, env_setup.py:281)."""
    torch = torch_cuda
    L, T = 8, 29
    g = torch.Generator(device='cuda').manual_seed(3)
    acts = torch.randint(0, B + 1, (T, E, U), generator=g, device='cuda', dtype=torch.uint8)
    ref = _make(kind, U, B, E, rand_episodes=rand, L=L)
    ref.reset()
    want_obs, want_rew = [], []
    for t in range(T):
        if ref.time == L:
            ref.reset()
        ref.step(acts[t])
        want_obs.append(ref.obs.clone()); want_rew.append(ref.reward.clone())
    env = _make(kind, U, B, E, rand_episodes=rand, L=L)
    env.reset()
    out = {'obs': torch.empty((T,) + tuple(env.obs.shape), device='cuda'), 'reward': torch.empty((T,) + tuple(env.reward.shape), device='cuda')}
    env.rollout(acts, out=out, horizon=L)
    env.check()
    assert torch.equal(out['obs'], torch.stack(want_obs)) and torch.equal(out['reward'], torch.stack(want_rew))
    assert env.time == ref.time and env.episode == ref.episode
    assert _same_state(_state(env), _state(ref))
    env.step(acts[0]); ref.step(acts[0])                     # and the handles agree on what comes next
    assert torch.equal(env.obs, ref.obs)


def test_rollout_reference_tape_fixed_episodes(torch_cuda):
    """rng='reference' with the reference's default (re-seeded at every reset, base.py:171-173): the borrowed tape is replayed."""
    torch = torch_cuda
    L, T, U, B, E = 6, 20, 5, 4, 9
    acts = torch.randint(0, B + 1, (T, E, U), device='cuda', dtype=torch.uint8)
    ref = _make('multi', U, B, E, rand_episodes=False, L=L, rng='reference')
    env = _make('multi', U, B, E, rand_episodes=False, L=L, rng='reference')
    ref.reset(); env.reset()
    for t in range(T):
        if ref.time == L:
            ref.reset()
        ref.step(acts[t])
    env.rollout(acts, horizon=L)
    env.check()
    assert torch.equal(env.obs, ref.obs) and _same_state(_state(env), _state(ref))
    # streams that continue across episodes (rand_episodes): the host draws every episode's tape, rollout() cuts itself at the
    # episode boundaries (round 2 raised NotImplementedError here)
    ref = _make('multi', U, B, E, rand_episodes=True, L=L, rng='reference')
    env = _make('multi', U, B, E, rand_episodes=True, L=L, rng='reference')
    ref.reset(); env.reset()
    want = torch.zeros((T,) + tuple(ref.obs.shape), device='cuda')
    for t in range(T):
        if ref.time == L:
            ref.reset()
        ref.step(acts[t])
        want[t] = ref.obs
    out = {'obs': torch.zeros_like(want), 'reward': torch.zeros((T,) + tuple(env.reward.shape), device='cuda')}
    env.rollout(acts[:7], out={k: v[:7] for k, v in out.items()}, horizon=L)          # two calls: the boundary logic carries over
    env.rollout(acts[7:], out={k: v[7:] for k, v in out.items()}, horizon=L)
    env.check()
    assert torch.equal(out['obs'], want) and _same_state(_state(env), _state(ref)) and env.time == ref.time


@pytest.mark.parametrize('tight', [False, True])
def test_fused_rollout_against_the_oracle(torch_cuda, tight, monkeypatch):
    """The fused kernel directly against the CPU oracle (not only against single steps): 4 096 x 10 x 5 central -- BASELINE
    config 2 -- 40 steps with a reset at the horizon, every step's observation and reward compared.  tight: the tightly
    packed instantiation (rollout_kernel_tight: six 10-UE envs per wavefront) that throughput-bound central batches take."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    from oracle import oracle as orc
    E, U, B, L, T = 4096, 10, 5, 25, 40
    scn = scenarios.grid_map(B, 'mixed').with_ues(num_slow=U)
    m, bs, ues = build_from_scenario(scn)
    monkeypatch.setenv('DCOMP_TIGHT', '1' if tight else '0')
    env = BatchedMobileEnv(m, bs, ues, 'central', num_envs=E, seed=42, rng='philox', rand_episodes=True, episode_length=L)
    assert env.fused_rollout and env.lanes_per_env == (U if tight else 16)
    oenvs = []
    for e in range(E):
        o = orc.OracleEnv(int(scn.width), int(scn.height), scn.bs_pos, scn.bs_sharing, ['slow'] * U, kind=orc.CENTRAL)
        o.set_philox(42, e)
        oenvs.append(o)
    ob = orc.OracleBatch(oenvs)
    rng = np.random.default_rng(0)
    a = rng.integers(0, B + 1, size=(T, E, U)).astype(np.uint8)
    env.reset(); ob.reset()
    out = {'obs': torch.empty((T, E, U * (2 * B + 1)), device='cuda'), 'reward': torch.empty((T, E), device='cuda'),
           'ue_dr': torch.empty((T, E, U), device='cuda'), 'ue_utility': torch.empty((T, E, U), device='cuda')}
    env.rollout(torch.from_numpy(a).cuda(), out=out, horizon=L)
    env.check()
    got_obs, got_rew = out['obs'].cpu().numpy(), out['reward'].cpu().numpy()
    got_dr, got_ut = out['ue_dr'].cpu().numpy(), out['ue_utility'].cpu().numpy()
    episode = 0
    for t in range(T):
        if t and t % L == 0:
            episode += 1
            for o in ob.envs:
                o.set_episode(episode)
            ob.reset()
        o_obs, o_rew, o_conn, o_pos = ob.step(a[t])
        # every step's per-UE data rate and relative-SNR block at 1e-5 RELATIVE against the oracle's FP64 values (tests/parity.py)
        r = parity.assert_rates(env, ob, f'step {t}', ue_dr=got_dr[t], ue_utility=got_ut[t], ewma=False)
        parity.assert_obs(got_obs[t], o_obs, 'central', U, B, dr_rel=r['dr_rel'], msg=f'step {t}')
        np.testing.assert_allclose(got_rew[t], o_rew, rtol=0, atol=1e-5, err_msg=f'step {t}')
    st = env.state_host()
    assert np.array_equal(st['pos'], o_pos) and np.array_equal(st['conn'], o_conn)
    parity.assert_rates(env, ob, 'final state', ue_dr=got_dr[-1], ue_utility=got_ut[-1])          # incl. the EWMA the kernel stored


@pytest.mark.parametrize('kind,rng,L', [('multi', 'philox', 30), ('central', 'philox', 30), ('multi', 'reference', 0), ('central', 'philox', 0),
                                        ('central', 'reference', 30)])
def test_rollout_with_ue_arrival_and_departure(torch_cuda, kind, rng, L):
    """rollout() of an env whose UE list changes (base.py:433-443, 592-618): the schedule is fed to dcomp_rollout_ex per step
    (counts; in rng='reference' mode also the host-drawn list positions / border points), resets at the horizon included.
    Bit-identical to `if time == L: reset(); step()` issued one call at a time -- every step's outputs, ids, state."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    E, T = 40, 70 if L else 28
    arrival = {2: 3, 5: -2, 9: 4, 14: -3, 20: 2, 21: 2, 26: -4}
    scn = scenarios.large_map('mixed').with_ues(num_static=1, num_slow=3, num_fast=2)
    m, bs, ues = build_from_scenario(scn)
    B = len(bs)
    mk = lambda: BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=21, rng=rng, rand_episodes=(rng == 'philox'), episode_length=L or 30,
                                  ue_arrival=arrival)
    ref, env = mk(), mk()
    U = ref.U
    g = torch.Generator(device='cuda').manual_seed(5)
    acts = torch.randint(0, B + 1, (T, E, U), generator=g, device='cuda', dtype=torch.uint8)
    keys = ('obs', 'reward', 'sum_utility', 'ue_dr', 'ue_utility')
    want = {k: [] for k in keys}
    ref.reset()
    for t in range(T):
        if L and ref.time == L:
            ref.reset()
        ref.step(acts[t])
        for k in keys:
            want[k].append(getattr(ref, k).clone())
    ref.check()
    want = {k: torch.stack(v) for k, v in want.items()}
    env.reset()
    out = {k: torch.full_like(v, float('nan')) for k, v in want.items()}
    cut = 11
    env.rollout(acts[:cut], out={k: v[:cut] for k, v in out.items()}, horizon=L or None)      # two fragments: events and state carry over
    env.rollout(acts[cut:], out={k: v[cut:] for k, v in out.items()}, horizon=L or None)
    env.check()
    for k in keys:
        assert torch.equal(out[k], want[k]), k
    assert env.time == ref.time and env.episode == ref.episode and env.num_ue == ref.num_ue
    for k in ('pos', 'mv', 'conn', 'ewma', 'uid'):
        assert torch.equal(getattr(env, k), getattr(ref, k)), k
    last = mk()                                              # last-step outputs only
    last.reset()
    last.rollout(acts[:cut])
    assert torch.equal(last.obs, want['obs'][cut - 1]) and torch.equal(last.reward, want['reward'][cut - 1])


@pytest.mark.parametrize('U,B,E,sharing', [(10, 5, 333, 'mixed'), (5, 3, 500, 'mixed'), (20, 8, 100, 'rate-fair'), (9, 4, 77, 'proportional-fair'),
                                           (17, 7, 40, 'mixed'), (21, 2, 64, 'resource-fair')])
def test_tight_rollout_equals_tight_steps(torch_cuda, U, B, E, sharing, monkeypatch):
    """rollout_kernel_tight (central envs packed G = U lanes each, the fused kernel of throughput-bound batches) against the
    tightly packed step kernel launched once per step -- the same packing and summation order: every step's outputs and the
    final state, resets at the horizon included; positions / masks / movement words bit-identical, floats to the last bits."""
    torch = torch_cuda
    monkeypatch.setenv('DCOMP_TIGHT', '1')
    L, T = 9, 23
    ref = _make('central', U, B, E, sharing=sharing, L=L)
    env = _make('central', U, B, E, sharing=sharing, L=L)
    assert env.lanes_per_env == U and env.fused_rollout
    g = torch.Generator(device='cuda').manual_seed(11)
    acts = torch.randint(0, B + 1, (T, E, U), generator=g, device='cuda', dtype=torch.uint8)
    keys = ('obs', 'reward', 'sum_utility', 'ue_dr', 'ue_utility')
    want = {k: [] for k in keys}
    ref.reset(); env.reset()
    for t in range(T):
        if ref.time == L:
            ref.reset()
        ref.step(acts[t])
        for k in keys:
            want[k].append(getattr(ref, k).clone())
    want = {k: torch.stack(v) for k, v in want.items()}
    out = {k: torch.full_like(v, float('nan')) for k, v in want.items()}
    env.rollout(acts[:6], out={k: v[:6] for k, v in out.items()}, horizon=L)
    env.rollout(acts[6:], out={k: v[6:] for k, v in out.items()}, horizon=L)
    env.check(); ref.check()
    for k in ('pos', 'mv', 'conn'):
        assert torch.equal(getattr(env, k), getattr(ref, k)), k
    assert env.time == ref.time and env.episode == ref.episode
    for k in keys:
        torch.testing.assert_close(out[k], want[k], rtol=2e-6, atol=2e-6, msg=lambda m_, k=k: f'{k}: {m_}')
    torch.testing.assert_close(env.ewma, ref.ewma, rtol=2e-6, atol=1e-30)


def test_rollout_argument_validation(torch_cuda):
    """Raw pointers cross the ABI: wrong dtype / device / size must be refused on the host."""
    torch = torch_cuda
    env = _make('multi', 4, 3, 8)
    env.reset()
    good = torch.zeros((5, 8, 4), dtype=torch.uint8, device='cuda')
    env.rollout(good)
    for bad in (good.to(torch.int32), good.cpu(), good[:, :, :3], good[:, ::2]):
        with pytest.raises(ValueError):
            env.rollout(bad)
    with pytest.raises(ValueError):
        env.rollout(good, out={'obs': torch.zeros((4, 8, 4, 13), device='cuda'), 'reward': torch.zeros((5, 8, 4), device='cuda')})
    with pytest.raises(ValueError):
        env.step_into(good[0], torch.zeros((8, 4, 12), device='cuda'), torch.zeros((8, 4), device='cuda'))
    with pytest.raises(ValueError):
        env.step_into(good[0].to(torch.int64), env.obs, env.reward)


@pytest.mark.parametrize('reward', ['avg', 'sum'])
def test_long_rollouts_of_multi_agent_envs_with_up_to_three_stations_are_fused(torch_cuda, reward):
    """Multi-agent rows are 4B + 1 floats per UE: up to three stations (the reference's stock small / medium maps) they stream
    well enough from registers that rollouts of >= 4 steps take the fused kernel at any batch size (65 536 x 5 x 3: 8.1 vs 12.1 us
    per step).  Same masks / positions as step(); floats to the last bits (step() packs this batch tightly)."""
    torch = torch_cuda
    E, U, B, T, L = 30000, 5, 3, 14, 9
    big, ref = _make('multi', U, B, E, reward=reward, L=L), _make('multi', U, B, E, reward=reward, L=L)
    assert big.fused_rollout and not _make('multi', U, 4, E).fused_rollout
    g = torch.Generator(device='cuda').manual_seed(4)
    acts = torch.randint(0, B + 1, (T, E, U), generator=g, device='cuda', dtype=torch.uint8)
    big.reset(); ref.reset()
    out = {'obs': torch.empty((T,) + tuple(big.obs.shape), device='cuda'), 'reward': torch.empty((T, E, U), device='cuda')}
    big.rollout(acts, out=out, horizon=L)
    for t in range(T):
        if ref.time == L:
            ref.reset()
        ref.step(acts[t])
        torch.testing.assert_close(out['obs'][t], ref.obs, rtol=2e-6, atol=2e-6)
        torch.testing.assert_close(out['reward'][t], ref.reward, rtol=0, atol=2e-5 * (U if reward == 'sum' else 1))
    big.check(); ref.check()
    assert torch.equal(big.pos, ref.pos) and torch.equal(big.conn, ref.conn) and torch.equal(big.mv, ref.mv) and big.time == ref.time
    torch.testing.assert_close(big.ewma, ref.ewma, rtol=1e-5, atol=1e-30)


def test_long_rollouts_of_small_central_envs_are_fused_at_any_batch_size(torch_cuda):
    """Central envs of <= 8 stations go through the fused kernel for rollouts of >= 4 steps however large the batch (short rows
    stream well from registers and the kernel boundary is a quarter of such a step); shorter rollouts and multi-agent envs of
    that size keep one launch per step.  Same masks / positions as step(); a batch of this size is packed tightly by step() AND by
    the fused kernel (rollout_kernel_tight, picked by the library's own dispatch here), floats to the last bits (<= 2e-6)."""
    torch = torch_cuda
    E, U, B, T = 20000, 10, 5, 12
    big = _make('central', U, B, E)
    assert big.fused_rollout and big.lanes_per_env == U              # step(): tight packing; rollout(): fused, tightly packed too
    multi = _make('multi', U, B, E)
    assert not multi.fused_rollout
    ref = _make('central', U, B, E)
    g = torch.Generator(device='cuda').manual_seed(4)
    acts = torch.randint(0, B + 1, (T, E, U), generator=g, device='cuda', dtype=torch.uint8)
    big.reset(); ref.reset()
    out = {'obs': torch.empty((T,) + tuple(big.obs.shape), device='cuda'), 'reward': torch.empty((T, E), device='cuda')}
    big.rollout(acts, out=out)
    for t in range(T):
        ref.step(acts[t])
        torch.testing.assert_close(out['obs'][t], ref.obs, rtol=2e-6, atol=2e-6)
        torch.testing.assert_close(out['reward'][t], ref.reward, rtol=0, atol=2e-6)
    big.check(); ref.check()
    assert torch.equal(big.pos, ref.pos) and torch.equal(big.conn, ref.conn) and torch.equal(big.mv, ref.mv)
    torch.testing.assert_close(big.ewma, ref.ewma, rtol=1e-5, atol=1e-30)
    big.rollout(acts[:2])                                            # a 2-step rollout: one (tightly packed) launch per step, like step()
    ref.step(acts[0]); ref.step(acts[1])
    torch.testing.assert_close(big.obs, ref.obs, rtol=2e-6, atol=2e-6)   # (the EWMA history differs in its last bits; masks must not)
    assert torch.equal(big.conn, ref.conn) and torch.equal(big.pos, ref.pos)


def test_rollout_with_a_horizon_after_a_live_reseed(torch_cuda):
    """seed(s, immediate=True) on a rand_episodes=False env splices the new streams into the RUNNING episode's tape; the
    reference re-seeds with the configured seed at the next reset (base.py:171-173).  rollout(horizon=L) used to reset inside the
    kernel and replay the splice; it must take the reset() path.  Against the step-by-step sequence (which the reseed_* fixtures
    pin on the reference)."""
    torch = torch_cuda
    E, U, B, L, T = 5, 6, 4, 10, 27
    g = torch.Generator(device='cuda').manual_seed(3)
    acts = torch.randint(0, B + 1, (T, E, U), generator=g, device='cuda', dtype=torch.uint8)

    def start(kind):
        env = _make(kind, U, B, E, rng='reference', rand_episodes=False, L=L)
        env.reset()
        for t in range(4):
            env.step(acts[t])
        env.seed(977, immediate=True)
        return env
    for kind in ('multi', 'central'):
        a, b = start(kind), start(kind)
        want = []
        for t in range(4, T):
            if a.time == L:
                a.reset()
            a.step(acts[t])
            want.append((a.obs.clone(), a.reward.clone()))
        frag = {'obs': torch.empty((T - 4,) + tuple(b.obs.shape), device='cuda'), 'reward': torch.empty((T - 4,) + tuple(b.reward.shape), device='cuda')}
        b.rollout(acts[4:], out=frag, horizon=L)
        a.check(); b.check()
        for i, (o, r) in enumerate(want):
            assert torch.equal(frag['obs'][i], o) and torch.equal(frag['reward'][i], r), (kind, i)
        assert _same_state(_state(a), _state(b)) and a.time == b.time


def test_rollout_event_schedule_is_validated_before_the_first_launch(torch_cuda):
    """an invalid arrival / departure entry at step t of a rollout used to surface after steps 0..t-1 had been launched
    (env.time advanced).  The call now does all T steps or nothing."""
    import ctypes
    torch = torch_cuda
    from deepcomp_amd import _lib, scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    m, bs, ues = build_from_scenario(scenarios.medium_map('mixed').with_ues(num_slow=3))
    env = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=8, seed=1, rng='philox', episode_length=20, ue_arrival={3: 1, 6: -1})
    env.reset()
    T = 6
    acts = torch.zeros((T, 8, env.U), dtype=torch.uint8, device='cuda')
    n_rem = np.zeros(T, dtype=np.int32); n_add = np.zeros(T, dtype=np.int32)
    n_add[1] = 1                     # fine: 3 -> 4 = max_ues
    n_add[4] = 1                     # 4 + 1 > max_ues: invalid, at step 4
    opts = _lib.DcompRolloutOpts(0, 0, 0, 0)
    opts.ev_n_remove, opts.ev_n_add = n_rem.ctypes.data, n_add.ctypes.data
    before = _state(env)
    rc = env._L.dcomp_rollout_ex(env._h, env._st_ref, ctypes.c_void_p(acts.data_ptr()), T, env._out_ref, ctypes.byref(opts), env._stream())
    assert rc == _lib.EINVAL and 'step 4' in _lib.last_error()
    torch.cuda.synchronize()
    assert env.time == 0 and env.num_ue == 3 and _same_state(before, _state(env))
    n_add[4] = 0
    _lib.check(env._L.dcomp_rollout_ex(env._h, env._st_ref, ctypes.c_void_p(acts.data_ptr()), T, env._out_ref, ctypes.byref(opts), env._stream()))
    assert env.time == T and env.num_ue == 4
    env.check()


def test_rollout_is_fused_depends_on_the_number_of_steps(torch_cuda):
    """`fused_rollout` is a property of the env; whether ONE rollout is one launch also depends on its length."""
    big = _make('central', 10, 5, 65536)
    assert big.fused_rollout and big.rollout_is_fused(50) and big.rollout_is_fused(4) and not big.rollout_is_fused(3)
    assert big.rollout_is_fused(1, policy_loop=True)
    small = _make('central', 10, 5, 4096)
    assert small.rollout_is_fused(1) and small.rollout_is_fused(100)
    assert not _make('multi', 128, 32, 64).rollout_is_fused(100)
    # an every-step fragment of >= 2^31 rows falls back to one launch per step instead of failing (64-bit offsets on the host)
    assert not big.rollout_is_fused(4096, every_step=True) and big.rollout_is_fused(4096, every_step=False)


@pytest.mark.parametrize('kind,U,B,E', [('multi', 32, 10, 64), ('central', 10, 5, 256)])
def test_fragments_beyond_the_fused_kernels_row_limit_fall_back_to_one_launch_per_step(torch_cuda, kind, U, B, E, monkeypatch):
    """an every-step fragment of >= 2^31 rows (num_steps x num_envs x num_ue) leaves the fused rollout kernel (32-bit row
    indices) for one launch per step with 64-bit offsets on the host -- a branch no test reached (it would take a 2^31-row
    fragment).  DCOMP_FUSED_ROW_LIMIT_LOG2 lowers the limit: the same rollout, fragment and final state, bit for bit; the
    closed loop included."""
    torch = torch_cuda
    T = 16
    g = torch.Generator(device='cuda').manual_seed(3)
    acts = torch.randint(0, B + 1, (T, E, U), generator=g, device='cuda', dtype=torch.uint8)
    fused = _make(kind, U, B, E)
    assert fused.rollout_is_fused(T, every_step=True)
    fused.reset()
    keys = ('obs', 'reward', 'sum_utility', 'ue_dr', 'ue_utility')
    want = {k: torch.empty((T,) + tuple(getattr(fused, k).shape), device='cuda') for k in keys}
    fused.rollout(acts, out=want)
    fused.check()
    monkeypatch.setenv('DCOMP_FUSED_ROW_LIMIT_LOG2', '12')                 # T * E * U = 32 768 resp. 40 960 rows >= 2^12
    env = _make(kind, U, B, E)
    assert not env.rollout_is_fused(T, every_step=True) and env.rollout_is_fused(T, every_step=False)
    env.reset()
    got = {k: torch.full_like(v, float('nan')) for k, v in want.items()}
    env.rollout(acts, out=got)
    env.check()
    for k in keys:
        assert torch.equal(got[k], want[k]), k
    assert _same_state(_state(env), _state(fused)) and env.time == T
    # the closed loop takes the same way out (round 6: one launch per step, each reading the actions the launch before wrote) -- against
    # the fused closed loop of the twin env, run with the limit back in place
    assert env.set_policy('fullcomp') and fused.set_policy('fullcomp')
    env.reset()
    got = {k: torch.full_like(v, float('nan')) for k, v in want.items()}
    env.rollout_policy(T, out=got, horizon=7)
    monkeypatch.delenv('DCOMP_FUSED_ROW_LIMIT_LOG2')
    fused.reset()
    fused.rollout_policy(T, out=want, horizon=7)
    for k in keys:
        assert torch.equal(got[k], want[k]), k
    assert _same_state(_state(env), _state(fused)) and env.time == fused.time == T - 14 and torch.equal(env.next_action, fused.next_action)
    env.check(); fused.check()
