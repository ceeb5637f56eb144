"""CPU-side checks of the product's host logic: the C-ABI library loads and exports every symbol of
include/dcomp.h, the in-library MT19937 reproduces stdlib ``random`` (the reference's generator), scenario
tables, spaces, config validation.  No GPU compute is called here."""
import ctypes
import os
import random
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from deepcomp_amd import build, _lib
    build.build()                      # hipcc cross-compiles gfx950 on a GPU-less host
    return _lib.load()


def test_abi_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(REPO, 'include', 'dcomp.h')).read()
    declared = sorted(set(re.findall(r'\b(dcomp_[a-z0-9_]+)\s*\(', hdr)))
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/dcomp.h but not exported by libdcomp_hip.so'
    assert b'gfx950' in lib.dcomp_version()


def test_ctypes_structs_list_the_headers_members_in_order():
    """The Python layer passes dcomp_state / dcomp_out / dcomp_tape by pointer: a member added to include/dcomp_types.h (round 4:
    dcomp_out.obs_compact) must appear in the ctypes mirror, in the same place."""
    import re
    from deepcomp_amd import _lib
    txt = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'dcomp_types.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    for cname, mirror in (('dcomp_state', _lib.DcompState), ('dcomp_out', _lib.DcompOut), ('dcomp_tape', _lib.DcompTape)):
        body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (cname, cname), txt, flags=re.S).group(1)
        members = [re.split(r'[\s*]+', m.strip())[-1] for decl in body.split(';') if decl.strip() for m in decl.split(',')]
        assert members == [f[0] for f in mirror._fields_], (cname, members)


def test_policy_entry_point_refuses_bad_arguments_on_the_host(lib):
    """dcomp_heuristic_actions validates before it touches the device (raw pointers cross the ABI): no GPU needed."""
    import ctypes
    from deepcomp_amd import _lib
    EINVAL = -1
    good = dict(policy=1, obs_kind=_lib.MULTI, num_envs=4, num_ue=3, num_bs=5, num_active=3, epsilon=0.5, cluster_mask=None)
    fake = ctypes.c_void_p(4096)                       # never dereferenced: every case below fails validation first
    assert lib.dcomp_heuristic_actions(None, fake, fake, None) == EINVAL
    for bad in (dict(policy=7), dict(obs_kind=5), dict(num_bs=65), dict(num_ue=0), dict(num_active=4), dict(policy=2, epsilon=1.5),      # (64 stations since round 5)
                dict(policy=3)):
        p = _lib.DcompPolicy(**{**good, **bad})
        assert lib.dcomp_heuristic_actions(ctypes.byref(p), fake, fake, None) == EINVAL, bad
        assert lib.dcomp_last_error()
    p = _lib.DcompPolicy(**good)
    assert lib.dcomp_heuristic_actions(ctypes.byref(p), None, fake, None) == EINVAL


def test_fragment_entry_points_on_the_host(lib):
    """dcomp_fragment_words is pure arithmetic; dcomp_pack_fragment / dcomp_unpack_fragment validate before they launch."""
    import ctypes
    EINVAL = -1
    assert lib.dcomp_fragment_words(32, 10) == 32 * 12 + 20                     # U (B + 2) + 2B words: 1 616 B instead of 5 248 B
    assert lib.dcomp_fragment_words(128, 32) * 4 == 17664
    assert lib.dcomp_fragment_words(1, 1) == 5 and lib.dcomp_fragment_words(256, 32) == 256 * 34 + 64
    assert lib.dcomp_fragment_words(32, 40) == 32 * 43 + 80 and lib.dcomp_fragment_words(1024, 64) == 1024 * 67 + 128    # round 6: two set words per UE above 32 stations
    for bad in ((0, 10), (1025, 10), (32, 0), (32, 65)):
        assert lib.dcomp_fragment_words(*bad) == -1
    fake = ctypes.c_void_p(4096)                       # never dereferenced: every case below fails validation first
    assert lib.dcomp_pack_fragment(None, 4, 32, 10, fake, fake, None) == EINVAL
    assert lib.dcomp_pack_fragment(fake, 4, 32, 10, fake, None, None) == EINVAL       # the flag word is not optional
    assert lib.dcomp_pack_fragment(fake, 0, 32, 10, fake, fake, None) == EINVAL
    assert lib.dcomp_pack_fragment(fake, 4, 32, 65, fake, fake, None) == EINVAL
    assert lib.dcomp_unpack_fragment(fake, 4, 1025, 10, fake, None) == EINVAL
    assert lib.dcomp_unpack_fragment(None, 4, 32, 10, fake, None) == EINVAL
    assert lib.dcomp_last_error()


def test_connect_threshold_matches_reference_constant(lib):
    # d_T(snr = 2e-8) = 68.92488308058013 m; "roughly 69 m" (station.py:9)
    assert lib.dcomp_connect_threshold() == pytest.approx(68.92488308058013, abs=1e-9)


def test_connect_boundary_is_the_reference_decision_to_the_last_bit(lib):
    """The range decision the kernels take is `fl(fl(dx*dx) + fl(dy*dy)) < X` (dcomp_connect_boundary_sq); the reference's is
    `snr(sqrt(dx*dx + dy*dy)) > 2e-8` (station.py:122-127, 222-226).  (1) d_T and X equal what tests/golden/gen_golden.py recorded from the
    reference's own Basestation methods, BIT for bit; (2) X is the smallest double whose correctly rounded root reaches d_T -- one ulp below
    fl(d_T * d_T), which round 5 compared against; (3) on 20 000 placements within doubles of the circle (tests/threshold_cases.py) `q < X` equals
    the oracle's literal can_connect(sqrt(q)); (4) the reference-run fixture's own can_connect column agrees."""
    import math
    from oracle import oracle as orc
    from tests import threshold_cases as tc
    g = np.load(os.path.join(REPO, 'tests', 'golden', 'traj_threshold_ulps_static_multi_s42.npz'))
    d_t, X = lib.dcomp_connect_threshold(), lib.dcomp_connect_boundary_sq()
    assert d_t == float(g['cfg_threshold_d']) == orc.connect_threshold_distance()
    assert X == float(g['cfg_threshold_q']) == tc.boundary_q(d_t)
    assert math.sqrt(X) >= d_t > math.sqrt(math.nextafter(X, 0.0))
    assert X == math.nextafter(d_t * d_t, 0.0), 'fl(d_T^2) is one ulp above the boundary for the reference constants'
    rng = np.random.default_rng(5)
    n_edge = 0
    for _ in range(20000):
        px, py = float(rng.integers(0, 600)), float(rng.integers(0, 600))
        theta = rng.uniform(0, 2 * math.pi) if rng.random() < 0.5 else rng.choice([0.0, math.pi]) + rng.uniform(2e-5, 0.03) * rng.choice([-1, 1])
        bx, by = tc.station_at_threshold(px, py, theta, int(rng.integers(-3, 4)), d_t, X, int(rng.integers(-2, 3)))
        q = tc.q_ref(px, py, bx, by)
        n_edge += int(abs(q - X) <= 2 * math.ulp(X))
        assert (q < X) == bool(orc.can_connect(math.sqrt(q))), (px, py, bx, by)
    assert n_edge > 5000
    bs = g['cfg_bs_pos']
    ue = g['cfg_ue_init_xy']
    for b, u in enumerate(g['placement_ue']):
        q = tc.q_ref(float(ue[u][0]), float(ue[u][1]), float(bs[b][0]), float(bs[b][1]))
        assert (q < X) == bool(g['placement_can_connect'][b])
    assert not g['placement_can_connect'][0] and tc.q_fused(0.0, 0.0, *bs[0]) < d_t * d_t       # VERDICT r5's placement: round 5 connected it


def _cfg(U, w, h, vlo, vhi, ix=None, iy=None):
    from deepcomp_amd import _lib
    c = _lib.DcompCfg()
    c.num_envs, c.num_ue, c.num_bs, c.map_w, c.map_h = 1, U, 1, w, h
    keep = [np.asarray(vlo, np.int32), np.asarray(vhi, np.int32)]
    c.ue_vel_lo, c.ue_vel_hi = keep[0].ctypes.data_as(_lib._ip), keep[1].ctypes.data_as(_lib._ip)
    if ix is not None:
        keep += [np.asarray(ix, np.int32), np.asarray(iy, np.int32)]
        c.ue_init_x, c.ue_init_y = keep[2].ctypes.data_as(_lib._ip), keep[3].ctypes.data_as(_lib._ip)
    return c, keep


@pytest.mark.parametrize('seeds', [[42, 43, 20042], [0, 1, 2 ** 31 - 1, 2 ** 40 + 17, -5]])
def test_mt_tape_equals_stdlib_random(lib, seeds):
    """dcomp_mt_draw_tape == random.Random(seed + 100*(i+1)).randint(...) in the reference's draw order
    (base.py:138-143, user.py:94-109, movement.py:110-130)."""
    from deepcomp_amd import rng
    U, w, h, depth = 5, 194, 120, 9
    vel = ['slow', 'fast', 0, 4, 'slow']
    vr = [rng.vel_range(v) for v in vel]
    c, keep = _cfg(U, w, h, [r[0] for r in vr], [r[1] for r in vr])
    pos0, trip = rng.mt_tape(c, seeds, depth)
    for e, s in enumerate(seeds):
        for i in range(U):
            pr, mr = random.Random(s + 100 * (i + 1)), random.Random(s + 100 * (i + 1))
            assert list(pos0[e * U + i]) == [pr.randint(0, w), pr.randint(0, h)]
            for k in range(depth):
                lo, hi = vr[i]
                want = [mr.randint(lo, hi) if lo != hi else lo, mr.randint(10, w - 10), mr.randint(10, h - 10), 0]
                assert list(trip[e * U + i, k]) == want
    # the stdlib-stream producer (rand_episodes=True path) hands out the same first episode
    st = rng.StdlibStreams(seeds, w, h, vel, [(-1, -1)] * U, depth)
    p2, t2 = st.draw_episode(reseed=False)
    assert np.array_equal(p2, pos0) and np.array_equal(t2, trip)


def test_mt_tape_fixed_start_positions(lib):
    from deepcomp_amd import rng
    c, keep = _cfg(2, 150, 100, [1, 0], [3, 0], ix=[-1, 77], iy=[-1, 5])
    pos0, trip = rng.mt_tape(c, [42], 3)
    pr = random.Random(142)
    assert list(pos0[0]) == [pr.randint(0, 150), pr.randint(0, 100)]
    assert list(pos0[1]) == [77, 5]
    assert trip[1, 0, 0] == 0


def test_stdlib_streams_continue_across_episodes():
    """rand_episodes=True: the next episode continues the movement stream after the triples actually consumed."""
    from deepcomp_amd import rng
    st = rng.StdlibStreams([42], 194, 120, ['slow'], [(-1, -1)], 6)
    p0, t0 = st.draw_episode(reseed=False)
    p1, t1 = st.draw_episode(reseed=False, consumed=np.array([2]))
    ref = random.Random(142)
    seq = [[ref.randint(1, 3), ref.randint(10, 184), ref.randint(10, 110)] for _ in range(8)]
    assert t0[0, :, :3].tolist() == seq[:6]
    assert t1[0, :, :3].tolist() == seq[2:8]
    pr = random.Random(142)
    first = [pr.randint(0, 194), pr.randint(0, 120)]
    second = [pr.randint(0, 194), pr.randint(0, 120)]
    assert p0[0].tolist() == first and p1[0].tolist() == second


def test_live_reseed_splices_the_new_streams_behind_each_cursor():
    """MobileEnv.seed() on a live env (base.py:132-143), host side of the product: both streams of every UE restart from the new
    seed, the running episode's tape continues behind each UE's cursor with the START of the new movement stream, and a
    rand_episodes=True env carries the new streams into the next episode (positions from the fresh position stream, movement
    from where the spliced stream was consumed)."""
    from deepcomp_amd import rng
    w, h, D = 194, 120, 7
    st = rng.StdlibStreams([42, 20042], w, h, ['slow', 5], [(-1, -1), (3, -1)], D, border=[10, 20])
    p0, t0 = st.draw_episode(reseed=False)
    cur = np.array([2, 1, 4, 3])                       # triples consumed so far, per (env, UE)
    _, t1 = st.reseed_live([977, 20977], cur)
    for e, base in enumerate((977, 20977)):
        for i, (vr, bb) in enumerate((((1, 3), 10), ((5, 5), 20))):
            row, c = e * 2 + i, int(cur[e * 2 + i])
            assert t1[row, :c].tolist() == t0[row, :c].tolist()                      # what was consumed stays
            r = random.Random(base + 100 * (i + 1))
            want = [[r.randint(*vr) if vr[0] != vr[1] else vr[0], r.randint(bb, w - bb), r.randint(bb, h - bb)] for _ in range(D - c + 3)]
            assert t1[row, c:, :3].tolist() == want[:D - c]
    # next episode: UE (0, 0) consumed 3 more triples of the spliced stream (cursor 5), the others none
    p2, t2 = st.draw_episode(reseed=False, consumed=np.array([5, 1, 4, 3]))
    r = random.Random(977 + 100)
    seq = [[r.randint(1, 3), r.randint(10, w - 10), r.randint(10, h - 10)] for _ in range(3 + D + 2)]
    assert t2[0, :, :3].tolist() == seq[3:3 + D]
    pr = random.Random(977 + 100)
    assert p2[0].tolist() == [pr.randint(0, w), pr.randint(0, h)]                     # the position stream started over too
    st.extend(D + 2)                                                                  # episodes that outlive the tape keep drawing
    assert st._trip[0, :, :3].tolist() == seq[3:3 + D + 2]


def test_live_reseed_with_a_changing_ue_list_matches_the_oracles_reading():
    """The product's DynamicStdlibStreams.reseed_live and the oracle's DynRefStreams / RefEventDraws (which the reference-run
    reseeddyn_* fixtures pin) splice the same tapes and restart the same event generators."""
    from deepcomp_amd import rng
    from oracle import oracle as orc
    w, h, D, U0, max_id = 300, 200, 9, 3, 6
    vel = ['slow', 'fast', 4]
    prod = rng.DynamicStdlibStreams([42], w, h, vel, [(-1, -1)] * U0, D, True, max_id)
    init, new, ev = orc.DynRefStreams(42, w, h, vel, depth=D, rand_episodes=True), orc.RefRngTape(42, w, h, ['slow'] * max_id, depth=D), \
        orc.RefEventDraws(42, w, h, rand_episodes=True)
    pp, pt = prod.draw_episode()
    op0, ot0 = init.draw_episode()
    _, ot1 = new.draw_episode()
    trip_all = np.concatenate([ot0, ot1])
    assert np.array_equal(pt[:, :, :3], trip_all) and np.array_equal(pp, op0)
    assert prod.departures(1, 3)[0].tolist() == ev.departures(1, 3) and prod.arrivals(2)[0].tolist() == [list(x) for x in ev.arrivals(2)]
    lists = [[(1, False), (3, False), (4, True), (5, True)]]                          # UE 2 has left, ids 4 and 5 arrived
    cursors = [[3, 2, 1, 2]]
    _, pt2 = prod.reseed_live([977], lists, cursors)
    ot2 = init.reseed_live(977, trip_all, [(u, b, c) for (u, b), c in zip(lists[0], cursors[0])])
    ev.reseed_live(977)
    assert np.array_equal(pt2[:, :, :3], ot2)
    assert prod.departures(2, 4)[0].tolist() == ev.departures(2, 4) and prod.arrivals(1)[0].tolist() == [list(x) for x in ev.arrivals(1)]
    _, pt3 = prod.extend(D + 4)                                                       # the spliced rows of arrived UEs survive an extension
    assert np.array_equal(pt3[:, :D, :3], ot2)
    r = random.Random(977 + 100 * 3)                                                  # id 4 sits at list position 3 (1-based): cursor 1
    want = [[r.randint(1, 3), r.randint(10, w - 10), r.randint(10, h - 10)] for _ in range(D + 3)]
    assert pt3[U0 + 3, 1:, :3].tolist() == want
    # next episode (streams continue): same draws on both sides
    consumed = [[5, 2, 4]]                                                            # UE 2 left at 2 triples (before the re-seed)
    end_list = [(1, False), (3, False), (5, True)]
    pp4, pt4 = prod.draw_episode([end_list], consumed)
    op4, ot4 = init.draw_episode(end_list, consumed[0])
    assert np.array_equal(pp4, op4) and np.array_equal(pt4[:U0, :D, :3], ot4)


def test_create_rejects_bad_config_without_gpu(lib):
    from deepcomp_amd import _lib
    c, keep = _cfg(3, 10, 10, [1, 1, 1], [3, 3, 3])          # map too small for the 10 m waypoint border
    h = ctypes.c_void_p()
    assert _lib.create(c, h) == _lib.EINVAL
    assert b'map' in lib.dcomp_last_error()
    with pytest.raises(ValueError):
        _lib.check(_lib.EINVAL)
    with pytest.raises(AssertionError):
        _lib.check(_lib.EACTION)


def test_abi_guard_rejects_a_stale_caller_without_gpu(lib):
    """dcomp_out grew a 7th pointer with nothing to stop a caller compiled against six.  Handles are now created through
    dcomp_create_v(version, struct sizes, ...): anything but the library's own values -> DCOMP_EABI, before the config is even looked at;
    the version-1 symbol `dcomp_create` refuses always."""
    from deepcomp_amd import _lib
    assert lib.dcomp_abi_version() == _lib.ABI_VERSION == 3 and b'ABI 3' in lib.dcomp_version()
    c, keep = _cfg(3, 100, 100, [1, 1, 1], [3, 3, 3])
    h = ctypes.c_void_p(1234)
    sizes = [ctypes.sizeof(x) for x in (_lib.DcompCfg, _lib.DcompState, _lib.DcompOut, _lib.DcompRolloutOpts)]
    assert lib.dcomp_create(ctypes.byref(c), ctypes.byref(h)) == _lib.EABI and h.value is None        # the ABI-1 entry point
    assert b'ABI-1' in lib.dcomp_last_error()
    stale_out = sizes[2] - ctypes.sizeof(ctypes.c_void_p)                                               # dcomp_out without obs_compact
    assert lib.dcomp_create_v(3, sizes[0], sizes[1], stale_out, sizes[3], ctypes.byref(c), ctypes.byref(h)) == _lib.EABI
    assert str(stale_out).encode() in lib.dcomp_last_error() and str(sizes[2]).encode() in lib.dcomp_last_error()
    stale_state = sizes[1] - ctypes.sizeof(ctypes.c_void_p)                                             # dcomp_state without conn_hi (ABI 2)
    assert lib.dcomp_create_v(3, sizes[0], stale_state, sizes[2], sizes[3], ctypes.byref(c), ctypes.byref(h)) == _lib.EABI
    assert lib.dcomp_create_v(2, *sizes, ctypes.byref(c), ctypes.byref(h)) == _lib.EABI
    c65, keep65 = _cfg(3, 100, 100, [1, 1, 1], [3, 3, 3])
    c65.num_bs = 65                                    # one station more than the two mask words hold
    assert _lib.create(c65, h) == _lib.EINVAL and b'num_bs<=64' in lib.dcomp_last_error()
    with pytest.raises(ImportError):
        _lib.check(_lib.EABI)
    # the header's own macro passes exactly these sizes: compile a two-line C caller against include/ and compare
    import subprocess
    import tempfile
    src = ('#include "dcomp.h"\n#include <stdio.h>\nint main(void){printf("%d %zu %zu %zu %zu\\n", DCOMP_ABI_VERSION, sizeof(dcomp_cfg), '
           'sizeof(dcomp_state), sizeof(dcomp_out), sizeof(dcomp_rollout_opts));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 'a.c'), 'w').write(src)
        subprocess.run(['gcc', '-I', os.path.join(REPO, 'include'), '-o', os.path.join(d, 'a'), os.path.join(d, 'a.c')], check=True)
        got = subprocess.run([os.path.join(d, 'a')], stdout=subprocess.PIPE, text=True, check=True).stdout.split()
    assert [int(x) for x in got] == [_lib.ABI_VERSION] + sizes, 'the ctypes mirrors and include/dcomp*.h disagree about a struct size'


def test_header_is_c99_and_the_c_host_example_builds_without_gpu(lib, tmp_path):
    """include/dcomp.h is a C header (the FFI of a Go / Java / C host binds it): it must pass a pedantic C99 compiler, and
    examples/c_host_step.cpp -- the non-Python host tests/test_c_host_gpu.py runs on the GPU -- must compile and link against the library."""
    import subprocess
    src = tmp_path / 'a.c'
    src.write_text('#include "dcomp.h"\nint main(void) { dcomp_cfg c = {0}; dcomp_out o = {0}; dcomp_env *e = 0; (void)o; return dcomp_create(&c, &e) == DCOMP_EABI; }\n')
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-Werror', '-pedantic', '-fsyntax-only', '-I', os.path.join(REPO, 'include'), str(src)], check=True)
    csrc = os.path.join(REPO, 'deepcomp_amd', 'csrc')
    r = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O2', '-I', os.path.join(REPO, 'include'), '-o', str(tmp_path / 'c_host_step'),
                        os.path.join(REPO, 'examples', 'c_host_step.cpp'), '-L', csrc, '-ldcomp_hip', f'-Wl,-rpath,{csrc}'],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]


def test_named_ue_arrival_schedules_are_the_references():
    """env_setup.py:205-226: the five `--ue-arrival` schedules as data, against what the reference's own get_ue_arrival returned
    (tests/golden/ue_arrival_schedules.json, recorded by gen_golden.py) -- and against the one a reference-run trajectory used."""
    import json
    from deepcomp_amd import scenarios as S
    from deepcomp_amd import rng as _rng
    want = json.load(open(os.path.join(REPO, 'tests', 'golden', 'ue_arrival_schedules.json')))
    assert sorted(S.UE_ARRIVAL) == sorted(want) and len(want) == 5
    for name, pairs in want.items():
        got = S.get_ue_arrival(name)
        assert list(got.items()) == [tuple(p) for p in pairs], name                 # same steps, same counts, same order
        assert got is not S.UE_ARRIVAL[name]
    assert S.get_ue_arrival(None) is None
    with pytest.raises(AssertionError):
        S.get_ue_arrival('sometimes')
    g = np.load(os.path.join(REPO, 'tests', 'golden', 'dyn_medium_central_largeupdown_s42.npz'))
    assert dict(zip(g['cfg_arrival_t'].tolist(), g['cfg_arrival_n'].tolist())) == S.UE_ARRIVAL['largeupdown']
    # "large increase up to 12 (starting at 1)": the capacity the env derives (base.py:79-84) from the schedule
    assert _rng.max_num_ue(1, 100, S.get_ue_arrival('largeupdown'), None) == int(g['cfg_max_ues']) == 12


def test_scenario_tables_match_reference_geometry():
    """Numbers of env_setup.py:52-176 (also recorded in the golden fixtures by the reference run)."""
    from deepcomp_amd import scenarios as S
    g = np.load(os.path.join(REPO, 'tests', 'golden', 'traj_custom4x4_multi_s42.npz'))
    scn = S.custom_map('mixed')
    assert np.array_equal(np.array(scn.bs_pos, float), g['cfg_bs_pos'])
    assert [S.SHARING_MODELS.index(s) for s in scn.bs_sharing] == [0, 1, 3, 0] == list(g['cfg_bs_sharing'])
    med = S.medium_map('resource-fair')
    assert (int(med.width), int(med.height)) == (120, 106)                    # map.py:20-21 int() truncation
    assert med.bs_pos[2] == (60.0, 10 + np.sqrt(100 ** 2 - 50 ** 2))
    assert S.large_map('mixed').num_bs == 7 and S.large_map('mixed', num_bs=3).width == 125
    grid = S.grid_map(10)
    assert (grid.width, grid.height, grid.num_bs) == (400, 300, 10) and grid.bs_pos[4] == (50, 150)
    assert S.grid_map(32).width == 600 and S.grid_map(5).height == 200
    ues = S.small_map().with_ues(num_static=1, num_slow=2, num_fast=1).ue_specs
    assert [u['id'] for u in ues] == ['1', '2', '3', '4'] and [u['velocity'] for u in ues] == [0, 'slow', 'slow', 'fast']
    with pytest.raises(AssertionError):
        S.sharing_for_bs('best-effort', 0)


def test_spaces_and_entities():
    from deepcomp_amd import spaces
    from deepcomp_amd.entities import Basestation, Map, Point, RandomWaypoint, User
    md = spaces.MultiDiscrete([4, 4, 4])
    assert md.contains([0, 3, 1]) and not md.contains([4, 0, 0]) and not md.contains([0, 0])
    assert list(spaces.Dict({'b': spaces.Discrete(2), 'a': spaces.Discrete(3)}).spaces.keys()) == ['a', 'b']
    m = Map(120.0, 106.6)
    assert (m.width, m.height) == (120, 106)
    with pytest.raises(AssertionError):
        Basestation('A', Point(0, 0), 'nope')
    with pytest.raises(AssertionError):
        User('1', m, 0, 0, RandomWaypoint(m, 1), util_func='quadratic')


@pytest.mark.skipif(not os.path.isdir('/root/reference/deepcomp'), reason='needs the reference checkout (build container only)')
def test_parse_entities_accepts_the_references_own_objects():
    """Drop-in contract: the env constructors read the reference's unmodified Map / Basestation / User /
    RandomWaypoint objects (built here behind the third-party stand-ins of tests/golden/_ref_shims.py)."""
    import sys
    sys.path.insert(0, os.path.join(REPO, 'tests', 'golden'))
    sys.path.insert(0, '/root/reference')
    import _ref_shims
    _ref_shims.install()
    from shapely.geometry import Point
    from deepcomp.env.entities.map import Map
    from deepcomp.env.entities.station import Basestation
    from deepcomp.env.entities.user import User
    from deepcomp.env.util.movement import RandomWaypoint
    from deepcomp_amd.env import parse_entities
    m = Map(width=194, height=120.7)
    bs = [Basestation('A', Point(10, 60), 'resource-fair'), Basestation('B', Point(97.5, 10), 'proportional-fair')]
    ues = [User('1', m, 'random', 'random', RandomWaypoint(m, velocity='slow')),
           User('2', m, 30, 'random', RandomWaypoint(m, velocity='fast'), util_func='step', dr_req=2),
           User('3', m, 'random', 7, RandomWaypoint(m, velocity=0))]
    e = parse_entities(m, bs, ues)
    assert (e['map_w'], e['map_h']) == (194, 120)
    assert e['bs_x'].tolist() == [10.0, 97.5] and e['bs_sharing'].tolist() == [0, 3]
    assert e['ue_ids'] == ['1', '2', '3'] and e['ue_util'].tolist() == [0, 1, 0] and e['ue_dr_req'].tolist() == [1.0, 2.0, 1.0]
    assert e['vel_lo'].tolist() == [1, 5, 0] and e['vel_hi'].tolist() == [3, 10, 0]
    assert e['init_xy'] == [(-1, -1), (30, -1), (-1, 7)]


def test_fuzz_specs_are_plain_data():
    """tools/fuzz_parity.py: configurations are JSON-able data; build_case() turns them into entity objects (host only)."""
    import json
    import sys
    sys.path.insert(0, os.path.join(REPO, 'tools'))
    import fuzz_parity
    from deepcomp_amd.env import parse_entities
    rng = np.random.default_rng(0)
    kinds = set()
    for _ in range(60):
        spec = fuzz_parity.random_spec(rng)
        c = fuzz_parity.build_case(json.loads(json.dumps(spec)))
        e = parse_entities(c['m'], c['bs'], c['ues'])
        assert len(e['ue_ids']) == spec['U'] and e['bs_x'].shape == (spec['B'],) and (e['map_w'], e['map_h']) == (spec['w'], spec['h'])
        assert e['bs_sharing'].tolist() == [fuzz_parity.SHARING.index(s) for s in spec['sh']]
        kinds.add((spec['tape'], spec['arrival'] is not None))
    assert len(kinds) == 4                                  # Philox / reference draws x fixed / changing UE lists all occur
    frozen = json.load(open(os.path.join(REPO, 'tests', 'golden', 'fuzz_maxcap_near_ties.json')))
    assert [s['U'] for s in frozen] == [130, 70] and all('max-cap' in s['sh'] for s in frozen)
    fuzz_parity.build_case(frozen[0])


def test_bench_self_spawn_reports_a_dead_rank_instead_of_hanging():
    """`python bench.py --gpus 2` without a launcher starts its ranks itself (bench.py::self_spawn).  Without a GPU the ranks
    die at set-up: the parent must notice, take the survivors down and exit non-zero with a message -- not sit in a rendezvous."""
    import subprocess
    import sys
    import time
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip('needs a box WITHOUT a GPU (the GPU suite runs the real thing)')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    t0 = time.time()
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--backend', 'gloo', '--same-device', '--envs', '64', '--steps', '2',
                        '--warmup', '1', '--no-cpu-baseline', '--no-also', '--no-stream'], cwd=REPO, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and 'self-spawned ranks failed' in r.stderr, (r.stdout[-500:], r.stderr[-1500:])
    assert time.time() - t0 < 240
    assert not [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
