"""GPU tests of the lossless compact rollout fragment (dcomp_pack_fragment / dcomp_unpack_fragment, include/dcomp.h).

Bar: unpack(pack(rows)) is BIT-identical to the rows -- on observation rows the REFERENCE produced (tests/golden/estack_*,
traj_grid128x32*, dyn_* fixtures: RelNormEnv.get_ue_obs, single_ue/variants.py:271-305), on the HIP path's own tensors at
BASELINE sizes (config 3, one GPU's share of config 5), through UE arrival / departure (zero rows of unlisted slots), and the
packer must refuse -- loudly, by its flag word -- anything it could not restore."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _rows_from_fixture(g):
    """[T+1, U, 4B+1] float32 rows in RLlib's sorted-key order from a reference-run fixture (reset + every step)."""
    out = []
    for pre in ('reset', 'step'):
        c, d = g[f'{pre}_obs_connected'], g[f'{pre}_obs_dr']
        out.append(np.concatenate([c, d, g[f'{pre}_obs_ues_at_bs'], g[f'{pre}_obs_util_at_bs'], g[f'{pre}_obs_utility'][..., None]], axis=-1))
    return np.concatenate(out, axis=0).astype(np.float32)


def _bits(t):
    import torch
    return t.contiguous().view(torch.int32)


@pytest.mark.parametrize('pattern', ['estack_grid32x10_multi_e*.npz', 'traj_grid128x32_multi_s42.npz', 'traj_grid64x16_multi_s42.npz',
                                     'traj_custom4x4_multi_s4*.npz', 'traj_large8x7_multi_*.npz', 'dyn_*_multi_*.npz'])
def test_round_trip_on_rows_the_reference_produced(torch_cuda, pattern):
    torch = torch_cuda
    from deepcomp_amd.fragment import FragmentCodec
    files = sorted(glob.glob(os.path.join(GOLDEN, pattern)))
    assert files
    for f in files:
        rows = _rows_from_fixture(np.load(f))
        T, U, row = rows.shape
        B = (row - 1) // 4
        x = torch.from_numpy(rows).cuda()
        codec = FragmentCodec(U, B)
        packed = codec.pack(x)
        CW = B + 2 + (B > 32)               # dr[B] | utility | connection word(s): two above 32 stations (dyn_dense40_*; round 6)
        assert packed.shape == (T, U * CW + 2 * B) and packed.dtype == torch.int32
        y = codec.unpack(packed)
        codec.check()
        assert torch.equal(_bits(x), _bits(y)), os.path.basename(f)
        # the record holds what it says: dr and utility copied, connection bits, the per-env columns once
        p = packed.cpu().numpy().view(np.uint32).reshape(T, -1)
        per_ue = p[:, :U * CW].reshape(T, U, CW)
        assert np.array_equal(per_ue[..., :B].view(np.float32), rows[..., B:2 * B])
        assert np.array_equal(per_ue[..., B].view(np.float32), rows[..., 4 * B])
        want_bits = (rows[..., :B] != 0).astype(np.uint64) @ (np.uint64(1) << np.arange(B, dtype=np.uint64))
        got_bits = per_ue[..., B + 1].astype(np.uint64) | ((per_ue[..., B + 2].astype(np.uint64) << np.uint64(32)) if B > 32 else np.uint64(0))
        assert np.array_equal(got_bits, want_bits)
        assert np.array_equal(p[:, U * CW:].view(np.float32), rows[:, 0, 2 * B:4 * B])


@pytest.mark.parametrize('E,U,B,T', [(65536, 32, 10, 1), (4096, 128, 32, 1), (1024, 32, 10, 8), (333, 7, 3, 5), (77, 130, 6, 2), (50, 5, 32, 3),
                                      (9, 256, 32, 2), (4096, 10, 5, 4), (100, 1, 1, 3)])
def test_round_trip_on_the_hip_path_at_scale(torch_cuda, E, U, B, T):
    """The step kernels' own observation tensors, stepped into a [T, E, U, 4B+1] fragment buffer."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    from deepcomp_amd.fragment import FragmentCodec
    m, bs, ues = build_from_scenario(scenarios.grid_map(B, 'mixed').with_ues(num_slow=U - U // 4, num_fast=U // 4))
    env = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=3, rng='philox', rand_episodes=True)
    g = torch.Generator(device='cuda').manual_seed(5)
    frag = torch.empty((T,) + tuple(env.obs.shape), device='cuda')
    rew = torch.empty((T,) + tuple(env.reward.shape), device='cuda')
    env.reset()
    for t in range(6 + T):
        a = torch.randint(0, B + 1, (E, U), generator=g, device='cuda', dtype=torch.uint8)
        if t < 6:
            env.step(a)
        else:
            env.step_into(a, frag[t - 6], rew[t - 6])
    env.check()
    codec = FragmentCodec(U, B)
    packed = codec.pack(frag)
    back = codec.unpack(packed)
    codec.check()
    assert packed.shape == (T, E, codec.words) and back.shape == frag.shape
    assert torch.equal(_bits(frag), _bits(back))
    assert packed.numel() <= frag.numel()                              # (equal only at 1 UE x 1 station)
    if (U, B) in ((32, 10), (128, 32)):                                # the BASELINE shapes: 3.2x / 3.7x fewer bytes on the links
        assert packed.numel() * 3.2 <= frag.numel()
    # into caller-provided buffers
    p2, b2 = torch.empty_like(packed), torch.full_like(frag, 7.0)
    codec.pack(frag, out=p2); codec.unpack(p2, out=b2)
    assert torch.equal(p2, packed) and torch.equal(_bits(b2), _bits(frag))


def test_round_trip_with_unlisted_ue_slots(torch_cuda):
    """UE arrival / departure: slots beyond the current list are all-zero rows and must come back as such."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    from deepcomp_amd.fragment import FragmentCodec
    m, bs, ues = build_from_scenario(scenarios.large_map('mixed').with_ues(num_slow=4, num_fast=2))
    env = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=300, seed=11, episode_length=40, rng='philox', rand_episodes=True,
                           ue_arrival={3: 2, 9: -3, 15: 4, 22: -2, 31: 1})
    codec = FragmentCodec(env.U, env.B)
    g = torch.Generator(device='cuda').manual_seed(5)
    env.reset()
    seen_dead = False
    for t in range(38):
        env.step(torch.randint(0, env.B + 1, (300, env.U), generator=g, device='cuda', dtype=torch.uint8))
        back = codec.unpack(codec.pack(env.obs))
        assert torch.equal(_bits(env.obs), _bits(back)), t
        n = env.num_ue
        if n < env.U:
            seen_dead = True
            assert not env.obs[:, n:].any() and env.obs[:, :n, env.B:2 * env.B].amax(-1).eq(1).all()
    codec.check()
    assert seen_dead


def test_pack_refuses_what_it_could_not_restore(torch_cuda):
    torch = torch_cuda
    from deepcomp_amd.fragment import FragmentCodec, fragment_words
    U, B = 32, 10
    rows = _rows_from_fixture(np.load(os.path.join(GOLDEN, 'estack_grid32x10_multi_e0.npz')))
    codec = FragmentCodec(U, B)
    x = torch.from_numpy(rows).cuda()
    codec.pack(x); codec.check()
    bad = x.clone(); bad[3, 5, 2 * B + 1] += 0.25                 # a replica of ues_at_bs that differs from row 0's
    codec.pack(bad)
    with pytest.raises(ValueError, match='per-env columns'):
        codec.check()
    bad = x.clone(); bad[1, 0, 3] = 0.5                           # `connected` is 0 / 1
    codec.pack(bad)
    with pytest.raises(ValueError, match='connected'):
        codec.check()
    codec.pack(x); codec.check()                                  # the flag word was cleared
    assert fragment_words(32, 10) == 32 * 12 + 20 == 404 and fragment_words(128, 32) * 4 == 17664
    with pytest.raises(ValueError):
        fragment_words(0, 10)
    with pytest.raises(ValueError):
        codec.pack(x[..., :-1].contiguous())
    with pytest.raises(ValueError):
        codec.unpack(torch.zeros((4, 403), dtype=torch.int32, device='cuda'))


def test_rollout_gather_hands_rows_over_as_the_compact_record(torch_cuda):
    """RolloutGather(codec=...): the fragment's observations cross the collective as the compact record and come back from
    GatherHandle.wait() as the rows, bit for bit (single-rank RCCL communicator on this GPU; the N > 1 indexing is covered by
    tests/test_sharded_gloo.py)."""
    import torch.distributed as dist
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    from deepcomp_amd.fragment import FragmentCodec
    from deepcomp_amd.sharded import RolloutBuffer, RolloutGather
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29571')
    own = not dist.is_initialized()
    if own:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))      # RCCL, one rank
    try:
        E, U, B, T = 512, 32, 10, 4
        m, bs, ues = build_from_scenario(scenarios.grid_map(B, 'mixed').with_ues(num_slow=U))
        env = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=9, rng='philox')
        env.reset()
        buf = RolloutBuffer(env, T)
        g = torch.Generator(device='cuda').manual_seed(3)
        frag = buf.collect(lambda obs: torch.randint(0, B + 1, (E, U), generator=g, device='cuda', dtype=torch.uint8))
        codec = FragmentCodec(U, B)
        gather = RolloutGather(use_side_stream=False, codec=codec)
        sent = {'obs': frag['obs'], 'reward': frag['reward']}
        h = gather.all_gather_async(sent)
        got = h.wait()
        codec.check()
        assert got['obs_compact'].shape == (1, T, E, codec.words) and got['obs'].shape == (1, T, E, U, 4 * B + 1)
        assert torch.equal(got['obs'][0].view(torch.int32), frag['obs'].view(torch.int32))
        assert torch.equal(got['reward'][0], frag['reward'])
        # a fragment the steps wrote as the compact record themselves (rollout(out={'obs_compact': ...})) goes through unchanged and
        # comes back as rows too: the same rows a twin env produces
        rows_env = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=9, rng='philox')
        comp_env = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=9, rng='philox')
        acts = torch.randint(0, B + 1, (T, E, U), generator=g, device='cuda', dtype=torch.uint8)
        rows = {'obs': torch.empty((T,) + tuple(rows_env.obs.shape), device='cuda'), 'reward': torch.empty((T,) + tuple(rows_env.reward.shape), device='cuda')}
        comp = {'obs_compact': torch.empty((T, E, codec.words), dtype=torch.int32, device='cuda'), 'reward': torch.empty_like(rows['reward'])}
        rows_env.reset(); comp_env.reset()
        rows_env.rollout(acts, out=rows); comp_env.rollout(acts, out=comp)
        got2 = gather.all_gather_async(dict(comp)).wait()
        assert torch.equal(got2['obs'][0].view(torch.int32), rows['obs'].view(torch.int32))
        assert torch.equal(got2['reward'][0], rows['reward'])
        # "bit-identical" is checked on this path, not assumed -- the pack kernel's flag word travels with the fragment and
        # wait() raises when the record was not lossless (rows that are no observation rows); check_lossless=False leaves the word
        assert got['pack_flags'].view(-1).tolist() == [0]
        bad = frag['obs'].clone()
        bad[1, 7, 3, 2 * B + 4] += 0.25                                    # a per-env column that differs between the rows of an env
        with pytest.raises(ValueError, match='rank 0: per-env columns differ'):
            gather.all_gather_async({'obs': bad, 'reward': frag['reward']}).wait()
        h3 = gather.all_gather_async(sent)                                  # the flag word was the bad fragment's own: the next one is clean
        assert h3.wait()['pack_flags'].view(-1).tolist() == [0]
        loose = RolloutGather(use_side_stream=False, codec=codec, check_lossless=False)
        assert loose.all_gather_async({'obs': bad, 'reward': frag['reward']}).wait()['pack_flags'].view(-1).tolist() == [1]
        # reuse_buffers covers the unpacked rows too (the largest tensor of a hand-off): k alternating buffers, no allocation per hand-off
        ring = RolloutGather(use_side_stream=False, codec=codec, reuse_buffers=2)
        ptrs = [ring.all_gather_async(sent).wait()['obs'].data_ptr() for _ in range(5)]
        assert ptrs[0] == ptrs[2] == ptrs[4] and ptrs[1] == ptrs[3] and ptrs[0] != ptrs[1]
        last = ring.all_gather_async(sent).wait()
        assert torch.equal(last['obs'][0].view(torch.int32), frag['obs'].view(torch.int32))
    finally:
        if own:
            dist.destroy_process_group()


# ---- the step writes the compact record itself (dcomp_out.obs_compact)
def _twin_envs(E, U, B, sharing='mixed', reward='avg', seed=3, **kw):
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    m, bs, ues = build_from_scenario(scenarios.grid_map(B, sharing).with_ues(num_slow=U - U // 4, num_fast=U // 4))
    mk = lambda: BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=seed, rng='philox', rand_episodes=True, reward=reward, **kw)
    return mk(), mk()


COMPACT_SHAPES = [(256, 32, 10), (65536, 32, 10), (4096, 128, 32), (333, 7, 3), (77, 130, 6), (50, 5, 32), (9, 256, 32), (40, 70, 28), (25, 100, 12),
                  (32768, 10, 5), (100, 1, 1), (200, 1, 32), (150, 3, 32), (31, 64, 24), (64, 20, 21), (3, 200, 32)]


@pytest.mark.parametrize('E,U,B', COMPACT_SHAPES)
def test_step_writes_the_compact_record_itself(torch_cuda, E, U, B):
    """dcomp_out.obs_compact: unpack(what step_compact wrote) is bit-identical to the rows step() writes, and identical to
    pack(rows) word for word -- on every step kernel (narrow padded, tightly packed, multi-wave envs, wide) and at reset."""
    torch = torch_cuda
    from deepcomp_amd.fragment import FragmentCodec
    if B not in _lib_b_list():
        pytest.skip(f"development build without B = {B}")
    rows_env, comp_env = _twin_envs(E, U, B)
    codec = FragmentCodec(U, B)
    g = torch.Generator(device='cuda').manual_seed(11)
    packed = torch.full((E, codec.words), -1, dtype=torch.int32, device='cuda')
    rew = torch.empty_like(comp_env.reward)
    rows_env.reset()
    comp_env.reset_compact(packed)
    assert torch.equal(_bits(codec.unpack(packed)), _bits(rows_env.obs))
    assert torch.equal(codec.pack(rows_env.obs), packed)
    for t in range(12):
        a = torch.randint(0, B + 1, (E, U), generator=g, device='cuda', dtype=torch.uint8)
        rows_env.step(a)
        packed.fill_(-1)
        comp_env.step_compact(a, packed, rew)
        assert torch.equal(_bits(codec.unpack(packed)), _bits(rows_env.obs)), t
        assert torch.equal(codec.pack(rows_env.obs), packed), t
        assert torch.equal(_bits(rew), _bits(rows_env.reward)), t
    codec.check(); rows_env.check(); comp_env.check()
    for k in ('pos', 'mv', 'conn', 'ewma'):
        assert torch.equal(getattr(rows_env, k), getattr(comp_env, k)), k
    if comp_env.log_metrics:
        assert torch.equal(_bits(rows_env.sum_utility), _bits(comp_env.sum_utility))


def _lib_b_list():
    """All station counts, unless this is a development build of a few (DCOMP_BUILD_B, deepcomp_amd/build.py)."""
    import os
    dev = os.environ.get('DCOMP_BUILD_B')
    return [int(x) for x in dev.split(',')] if dev else list(range(1, 33))


@pytest.mark.parametrize('E,U,B,reward,sharing', [(128, 10, 5, 'min', 'resource-fair'), (64, 32, 10, 'sum', 'mixed'), (16, 128, 32, 'sum', 'mixed'),
                                                  (16, 128, 32, 'min', 'resource-fair'), (40, 9, 5, 'avg', 'max-cap')])
def test_compact_record_with_other_rewards_and_sharing_models(torch_cuda, E, U, B, reward, sharing):
    torch = torch_cuda
    from deepcomp_amd.fragment import FragmentCodec
    if B not in _lib_b_list():
        pytest.skip(f"development build without B = {B}")
    rows_env, comp_env = _twin_envs(E, U, B, sharing=sharing, reward=reward)
    codec = FragmentCodec(U, B)
    g = torch.Generator(device='cuda').manual_seed(12)
    packed = torch.empty((E, codec.words), dtype=torch.int32, device='cuda')
    rew = torch.empty_like(comp_env.reward)
    rows_env.reset(); comp_env.reset()
    for t in range(8):
        a = torch.randint(0, B + 1, (E, U), generator=g, device='cuda', dtype=torch.uint8)
        rows_env.step(a)
        comp_env.step_compact(a, packed, rew)
        assert torch.equal(_bits(codec.unpack(packed)), _bits(rows_env.obs)), t
        assert torch.equal(_bits(rew), _bits(rows_env.reward)), t


@pytest.mark.parametrize('E,U,B,T', [(4096, 10, 5, 7), (512, 32, 10, 6), (65536, 32, 10, 3), (64, 128, 32, 4), (300, 3, 2, 9)])
def test_rollout_writes_compact_fragments(torch_cuda, E, U, B, T):
    """rollout(out={'obs_compact': [T, E, words]}) -- fused (one launch, records straight from registers) and one launch per step --
    against the row fragment of a twin env; with a reset at the horizon inside the rollout."""
    torch = torch_cuda
    from deepcomp_amd.fragment import FragmentCodec
    if B not in _lib_b_list():
        pytest.skip(f"development build without B = {B}")
    rows_env, comp_env = _twin_envs(E, U, B, episode_length=5)
    codec = FragmentCodec(U, B)
    g = torch.Generator(device='cuda').manual_seed(13)
    acts = torch.randint(0, B + 1, (T, E, U), generator=g, device='cuda', dtype=torch.uint8)
    frag = {'obs': torch.empty((T,) + tuple(rows_env.obs.shape), device='cuda'), 'reward': torch.empty((T,) + tuple(rows_env.reward.shape), device='cuda')}
    cfrag = {'obs_compact': torch.full((T, E, codec.words), -1, dtype=torch.int32, device='cuda'), 'reward': torch.empty_like(frag['reward'])}
    rows_env.reset(); comp_env.reset()
    rows_env.rollout(acts, out=frag, horizon=5)
    comp_env.rollout(acts, out=cfrag, horizon=5)
    assert torch.equal(_bits(codec.unpack(cfrag['obs_compact'])), _bits(frag['obs']))
    assert torch.equal(codec.pack(frag['obs']), cfrag['obs_compact'])
    assert torch.equal(_bits(cfrag['reward']), _bits(frag['reward']))
    codec.check(); rows_env.check(); comp_env.check()
    assert torch.equal(rows_env.pos, comp_env.pos) and torch.equal(rows_env.conn, comp_env.conn)


def test_compact_record_is_refused_where_it_is_not_defined(torch_cuda):
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    m, bs, ues = build_from_scenario(scenarios.grid_map(5, 'mixed').with_ues(num_slow=4))
    central = BatchedMobileEnv(m, bs, ues, 'central', num_envs=8, seed=1, rng='philox')
    central.reset()
    a = torch.zeros((8, 4), dtype=torch.uint8, device='cuda')
    with pytest.raises(NotImplementedError):
        central.step_compact(a, torch.empty((8, 4 * 7 + 10), dtype=torch.int32, device='cuda'), central.reward)
    multi = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=8, seed=1, rng='philox')
    multi.reset()
    with pytest.raises(ValueError):                                        # wrong size / dtype never reaches the kernel
        multi.step_compact(a, torch.empty((8, 4 * 7 + 9), dtype=torch.int32, device='cuda'), multi.reward)
    with pytest.raises(ValueError):
        multi.step_compact(a, torch.empty((8, 4 * 7 + 10), dtype=torch.float32, device='cuda'), multi.reward)
    # the C ABI says the same for callers that bypass the Python checks
    import ctypes
    from deepcomp_amd import _lib
    out = _lib.DcompOut(central.obs.data_ptr(), central.reward.data_ptr(), None, None, None, None, central.obs.data_ptr())
    rc = _lib.load().dcomp_step(central._h, central._st_ref, ctypes.c_void_p(a.data_ptr()), ctypes.byref(out), central._stream())
    assert rc == _lib.EINVAL


@pytest.mark.parametrize('E,U,B,policy', [(64, 32, 10, '3gpp'), (4096, 10, 5, 'dynamic'), (16, 128, 32, 'fullcomp'), (24, 70, 24, '3gpp'), (2000, 32, 10, 'cluster'),
                                          (40, 32, 40, 'dynamic'), (3, 300, 12, '3gpp'), (9, 20, 64, 'fullcomp')])      # the generic kernel (round 6)
def test_closed_policy_loop_with_compact_records(torch_cuda, E, U, B, policy):
    """dcomp_set_policy decides on the registers the observation is written from, so a heuristic-driven loop needs no rows: twin
    envs, one stepping rows, one the compact record, take the same decisions and produce the same observations -- step by step and
    through rollout_policy (fused closed loop where the shape has one, one launch per step elsewhere), resets at the horizon included."""
    torch = torch_cuda
    from deepcomp_amd.fragment import FragmentCodec
    rows_env, comp_env = _twin_envs(E, U, B, episode_length=6)
    codec = FragmentCodec(U, B)
    kw = {}
    if policy == 'dynamic':
        kw['epsilon'] = 0.4
    if policy == 'cluster':
        kw['cluster_mask'] = torch.tensor([7 << (3 * (b // 3)) & ((1 << B) - 1) for b in range(B)], dtype=torch.int32, device='cuda')   # clusters of 3
    for env in (rows_env, comp_env):
        env.reset()
        env.set_policy(policy, **kw)
        env.reset()                                   # the first decision comes from the reset kernel
    packed = torch.empty((E, codec.words), dtype=torch.int32, device='cuda')
    rew = torch.empty_like(comp_env.reward)
    for t in range(5):
        assert torch.equal(rows_env.next_action, comp_env.next_action), t
        rows_env.step(rows_env.next_action)
        comp_env.step_compact(comp_env.next_action, packed, rew)
        assert torch.equal(_bits(codec.unpack(packed)), _bits(rows_env.obs)), t
    T = 9
    rows = {'obs': torch.empty((T,) + tuple(rows_env.obs.shape), device='cuda'), 'reward': torch.empty((T,) + tuple(rows_env.reward.shape), device='cuda')}
    comp = {'obs_compact': torch.empty((T, E, codec.words), dtype=torch.int32, device='cuda'), 'reward': torch.empty_like(rows['reward'])}
    rows_env.rollout_policy(T, out=rows, horizon=6)
    comp_env.rollout_policy(T, out=comp, horizon=6)
    assert torch.equal(_bits(codec.unpack(comp['obs_compact'])), _bits(rows['obs']))
    assert torch.equal(_bits(comp['reward']), _bits(rows['reward']))
    assert torch.equal(rows_env.next_action, comp_env.next_action)
    assert torch.equal(rows_env.pos, comp_env.pos) and torch.equal(rows_env.conn, comp_env.conn)
    rows_env.check(); comp_env.check()



@pytest.mark.parametrize('E,large,nslow,nfast,arrival', [(300, True, 4, 2, {3: 2, 9: -3, 15: 4, 22: -2, 31: 1}),
                                                          (64, False, 60, 10, {2: 5, 4: -20, 11: 30, 20: -3, 28: 40}),
                                                          (5000, True, 3, 0, {1: 1, 2: 1, 3: -2, 8: 3, 9: 3, 30: -5})])
def test_compact_record_with_ue_arrival_and_departure(torch_cuda, E, large, nslow, nfast, arrival):
    """A changing UE list: the record of an unlisted slot is all zeros, the per-env columns come from the env (the last slot's lane may
    be unlisted); twin envs as above, step by step and through rollout()'s event feed with a reset at the horizon."""
    torch = torch_cuda
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    from deepcomp_amd.fragment import FragmentCodec
    scn = (scenarios.large_map('mixed') if large else scenarios.grid_map(12, 'mixed')).with_ues(num_slow=nslow, num_fast=nfast)
    m, bs, ues = build_from_scenario(scn)
    mk = lambda: BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=11, episode_length=40, rng='philox', rand_episodes=True, ue_arrival=arrival)
    rows_env, comp_env = mk(), mk()
    U, B = rows_env.U, rows_env.B
    codec = FragmentCodec(U, B)
    g = torch.Generator(device='cuda').manual_seed(5)
    packed = torch.full((E, codec.words), -1, dtype=torch.int32, device='cuda')
    rew = torch.empty_like(comp_env.reward)
    rows_env.reset(); comp_env.reset_compact(packed)
    assert torch.equal(_bits(codec.unpack(packed)), _bits(rows_env.obs)) and torch.equal(codec.pack(rows_env.obs), packed)
    seen_dead = False
    for t in range(34):
        a = torch.randint(0, B + 1, (E, U), generator=g, device='cuda', dtype=torch.uint8)
        rows_env.step(a)
        packed.fill_(-1)
        comp_env.step_compact(a, packed, rew)
        assert torch.equal(_bits(codec.unpack(packed)), _bits(rows_env.obs)), t
        assert torch.equal(codec.pack(rows_env.obs), packed), t
        assert torch.equal(_bits(rew), _bits(rows_env.reward)), t
        assert rows_env.num_ue == comp_env.num_ue
        seen_dead |= rows_env.num_ue < U
    assert seen_dead
    T = 12                                                   # crosses the horizon (40): events start over with the new episode
    acts = torch.randint(0, B + 1, (T, E, U), generator=g, device='cuda', dtype=torch.uint8)
    rows = {'obs': torch.empty((T,) + tuple(rows_env.obs.shape), device='cuda'), 'reward': torch.empty((T,) + tuple(rows_env.reward.shape), device='cuda')}
    comp = {'obs_compact': torch.empty((T, E, codec.words), dtype=torch.int32, device='cuda'), 'reward': torch.empty_like(rows['reward'])}
    rows_env.rollout(acts, out=rows, horizon=40); comp_env.rollout(acts, out=comp, horizon=40)
    assert torch.equal(_bits(codec.unpack(comp['obs_compact'])), _bits(rows['obs']))
    assert torch.equal(codec.pack(rows['obs']), comp['obs_compact'])
    codec.check(); rows_env.check(); comp_env.check()
    assert torch.equal(rows_env.uid, comp_env.uid) and torch.equal(rows_env.pos, comp_env.pos)
