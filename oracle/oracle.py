"""ctypes front-end of the CPU ORACLE (``oracle/dcomp_oracle.c``).       *** TEST INFRASTRUCTURE ***

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this
module; it is the checker, never the thing measured or shipped.  Parity status: pinned against the
reference-generated fixtures in ``tests/golden`` by ``tests/test_oracle_golden.py``.

The reference draws start positions / velocities / waypoints from two per-UE ``random.Random``
streams (``base.py:132-143``, ``user.py:94-109``, ``movement.py:110-130``).
``RefRngTape`` reproduces that draw order with the very same stdlib generator and hands the oracle a
pre-drawn tape; the C side never needs Mersenne-Twister.
"""
import ctypes
import os
import random
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '_ref', 'libdcomp_oracle.so')

CENTRAL, MULTI = 0, 1
AVG, SUM, MIN = 0, 1, 2
RES_FAIR, RATE_FAIR, MAX_CAP, PROP_FAIR = 0, 1, 2, 3
SHARING_CODE = {'resource-fair': 0, 'rate-fair': 1, 'max-cap': 2, 'proportional-fair': 3}
UTIL_LOG, UTIL_STEP = 0, 1

_c_dp = ctypes.POINTER(ctypes.c_double)
_c_ip = ctypes.POINTER(ctypes.c_int32)
_c_fp = ctypes.POINTER(ctypes.c_float)
_c_u8p = ctypes.POINTER(ctypes.c_uint8)
_c_u32p = ctypes.POINTER(ctypes.c_uint32)


def build(force=False):
    """Compile the oracle into oracle/_ref/ (gcc, a few hundred ms)."""
    src = os.path.join(_HERE, 'dcomp_oracle.c')
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.orc_create.restype = ctypes.c_void_p
        L.orc_create.argtypes = [ctypes.c_int] * 6 + [_c_dp, _c_dp, _c_ip, _c_ip, _c_dp, _c_ip, _c_ip, _c_ip, _c_ip]
        L.orc_destroy.argtypes = [ctypes.c_void_p]
        L.orc_set_tape.argtypes = [ctypes.c_void_p, ctypes.c_int, _c_ip, _c_ip]
        L.orc_set_philox.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int64]
        L.orc_set_episode.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        L.orc_reset.argtypes = [ctypes.c_void_p]
        L.orc_step.argtypes = [ctypes.c_void_p, _c_ip]
        L.orc_step.restype = ctypes.c_int
        L.orc_get_obs.argtypes = [ctypes.c_void_p, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp]
        L.orc_get_reward.argtypes = [ctypes.c_void_p, _c_dp]
        L.orc_get_reward_before.argtypes = [ctypes.c_void_p, _c_dp]
        L.orc_get_state.argtypes = [ctypes.c_void_p, _c_dp, _c_dp, _c_dp, _c_ip, _c_ip, _c_u8p, _c_dp, _c_dp, _c_dp,
                                    _c_dp, _c_ip]
        L.orc_sum_utility.argtypes = [ctypes.c_void_p]
        L.orc_sum_utility.restype = ctypes.c_double
        L.orc_time.argtypes = [ctypes.c_void_p]
        L.orc_tape_cursor.argtypes = [ctypes.c_void_p, ctypes.c_int]
        for f in ('orc_snr', 'orc_dr_unshared', 'orc_log_utility'):
            getattr(L, f).argtypes = [ctypes.c_double]
            getattr(L, f).restype = ctypes.c_double
        L.orc_step_utility.argtypes = [ctypes.c_double, ctypes.c_double]
        L.orc_step_utility.restype = ctypes.c_double
        L.orc_can_connect.argtypes = [ctypes.c_double]
        L.orc_connect_threshold_distance.restype = ctypes.c_double
        L.orc_philox4x32_10.argtypes = [_c_u32p, _c_u32p, _c_u32p]
        L.orc_set_initial_ues.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_set_movement.argtypes = [ctypes.c_void_p, ctypes.c_int, _c_ip, _c_ip]
        L.orc_set_velocity.argtypes = [ctypes.c_void_p, ctypes.c_int, _c_dp]
        L.orc_probe_data_rate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _c_dp]
        L.orc_probe_data_rate.restype = ctypes.c_double
        L.orc_set_events.argtypes = [ctypes.c_void_p, ctypes.c_int, _c_ip, ctypes.c_int, _c_ip]
        L.orc_set_tape_ids.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _c_ip, _c_ip]
        L.orc_num_ue.argtypes = [ctypes.c_void_p]
        L.orc_get_uids.argtypes = [ctypes.c_void_p, _c_ip]
        L.orc_slot_born.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_get_orig_consumed.argtypes = [ctypes.c_void_p, _c_ip]
        L.orc_batch_reset.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _c_fp, ctypes.c_int]
        L.orc_batch_step.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _c_u8p, _c_fp, _c_fp, _c_u32p,
                                     _c_dp, ctypes.c_int]
        L.orc_batch_rates.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _c_dp, _c_dp, _c_dp, _c_dp, ctypes.c_int]
        L.orc_batch_conn_hi.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _c_u32p]
        L.orc_max_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _p(a, typ):
    return a.ctypes.data_as(typ) if a is not None else None


def vel_range(v):
    """Velocity spec -> inclusive integer draw range (movement.py:112-117)."""
    if v == 'slow' or v == -1:
        return 1, 3
    if v == 'fast' or v == -2:
        return 5, 10
    if float(v) != int(v):                 # a fixed velocity that is no integer (movement.py:116-117): never drawn, see vel_number
        return 0, 0
    return int(v), int(v)


def vel_number(v):
    """Fixed velocity given as a non-integer number (RandomWaypoint(map, velocity=2.5)) -> that number, else -1."""
    return float(v) if not isinstance(v, str) and float(v) >= 0 and float(v) != int(v) else -1.0


class RefRngTape:
    """Pre-draws what the reference's per-UE ``random.Random`` streams would hand out.

    seed rule: UE i (0-based) gets ``seed + 100*(i+1)`` for BOTH its position and its movement
    stream (base.py:138-143, user.py:94-96).  ``rand_episodes=False`` re-seeds at every reset
    (base.py:171-173); ``True`` lets the streams continue, which needs the number of triples the
    previous episode consumed (``consumed``)."""

    def __init__(self, seed, map_w, map_h, vel_specs, init_xy=None, depth=48, rand_episodes=False, border=None):
        self.seed, self.w, self.h, self.depth, self.rand_episodes = seed, int(map_w), int(map_h), depth, rand_episodes
        self.vel_specs = list(vel_specs)
        self.U = len(self.vel_specs)
        self.border = [10] * self.U if border is None else [int(b) for b in border]      # movement.py:87,126-127
        self.init_xy = init_xy if init_xy is not None else [(-1, -1)] * self.U
        self._seed_streams()
        self._states = None
        self._shift = None
        self._trip = None

    def _seed_streams(self, seed=None):
        seed = self.seed if seed is None else seed
        self.pos_rng = [random.Random(seed + 100 * (i + 1)) for i in range(self.U)]
        self.mov_rng = [random.Random(seed + 100 * (i + 1)) for i in range(self.U)]

    def reseed_live(self, seed, cursors=None):
        """MobileEnv.seed(seed) on a live env (base.py:132-143): BOTH streams of every UE start over from seed + 100*(i+1) at
        once.  What UE i still draws in the running episode comes from the start of its new movement stream: returns the
        episode's triples with row i continued from position cursors[i] (= triples consumed so far, OracleEnv.cursors()) by
        the new stream, or None when no episode has been drawn yet.  The configured seed stays what reset() of a
        rand_episodes=False env re-seeds with (base.py:171-173)."""
        self._seed_streams(seed)
        if self._trip is None:
            self._states = None
            return None
        trip = self._trip.copy()
        self._states, self._shift = [], [int(c) for c in cursors]
        for i in range(self.U):
            st = [self.mov_rng[i].getstate()]
            for k in range(self._shift[i], self.depth):
                trip[i, k] = self._triple(i)
                st.append(self.mov_rng[i].getstate())
            self._states.append(st)
        self._trip = trip
        return trip

    def _triple(self, i):
        r, v = self.mov_rng[i], self.vel_specs[i]
        lo, hi = vel_range(v)
        vel = r.randint(lo, hi) if lo != hi else lo      # only 'slow'/'fast' consume a draw
        bb = self.border[i]
        wx = r.randint(bb, int(self.w - bb))
        wy = r.randint(bb, int(self.h - bb))
        return vel, wx, wy

    def draw_episode(self, consumed=None):
        """Returns (pos0[U,2] int32, triples[U,depth,3] int32) for the next reset()."""
        if not self.rand_episodes:
            self._seed_streams()
        elif self._states is not None:
            assert consumed is not None, "rand_episodes=True needs the consumed-triple counts of the last episode"
            for i in range(self.U):
                self.mov_rng[i].setstate(self._states[i][int(consumed[i]) - (self._shift[i] if self._shift else 0)])
        self._shift = None
        pos0 = np.zeros((self.U, 2), dtype=np.int32)
        trip = np.zeros((self.U, self.depth, 3), dtype=np.int32)
        self._states = []
        for i in range(self.U):
            ix, iy = self.init_xy[i]
            pos0[i, 0] = self.pos_rng[i].randint(0, self.w) if ix < 0 else ix
            pos0[i, 1] = self.pos_rng[i].randint(0, self.h) if iy < 0 else iy
            st = [self.mov_rng[i].getstate()]
            for k in range(self.depth):
                trip[i, k] = self._triple(i)
                st.append(self.mov_rng[i].getstate())
            self._states.append(st)
        self._trip = trip
        return pos0, trip


class OracleEnv:
    """One env instance of the oracle."""

    def __init__(self, map_w, map_h, bs_pos, bs_sharing, vel_specs, kind=MULTI, reward_agg=AVG, ue_util=None,
                 ue_dr_req=None, init_xy=None, max_ues=None, pause=None, border=None):
        L = lib()
        self.U0 = len(vel_specs)                      # UEs in the configured ue_list
        self.U, self.B = max(max_ues or 0, self.U0), len(bs_pos)     # capacity (max_ues, base.py:79-84)
        self.kind = kind
        U, B = self.U, self.B
        pad = U - self.U0
        vel_specs = list(vel_specs) + ['slow'] * pad
        if ue_util is not None:
            ue_util = list(ue_util) + [0] * pad
        if ue_dr_req is not None:
            ue_dr_req = list(ue_dr_req) + [1.0] * pad
        if init_xy is not None:
            init_xy = [tuple(p) for p in init_xy] + [(-1, -1)] * pad
        bs = np.asarray(bs_pos, dtype=np.float64).reshape(B, 2)
        self._bx, self._by = np.ascontiguousarray(bs[:, 0]), np.ascontiguousarray(bs[:, 1])
        self._sh = np.asarray([SHARING_CODE.get(s, s) for s in bs_sharing], dtype=np.int32)
        self._util = np.zeros(U, dtype=np.int32) if ue_util is None else np.asarray(ue_util, dtype=np.int32)
        self._req = np.ones(U, dtype=np.float64) if ue_dr_req is None else np.asarray(ue_dr_req, dtype=np.float64)
        rng = [vel_range(v) for v in vel_specs]
        self._vlo = np.asarray([r[0] for r in rng], dtype=np.int32)
        self._vhi = np.asarray([r[1] for r in rng], dtype=np.int32)
        ixy = np.full((U, 2), -1, dtype=np.int32) if init_xy is None else np.asarray(init_xy, dtype=np.int32)
        self._ix, self._iy = np.ascontiguousarray(ixy[:, 0]), np.ascontiguousarray(ixy[:, 1])
        self.h = L.orc_create(U, B, int(map_w), int(map_h), kind, reward_agg, _p(self._bx, _c_dp), _p(self._by, _c_dp),
                              _p(self._sh, _c_ip), _p(self._util, _c_ip), _p(self._req, _c_dp), _p(self._vlo, _c_ip),
                              _p(self._vhi, _c_ip), _p(self._ix, _c_ip), _p(self._iy, _c_ip))
        self.h = ctypes.c_void_p(self.h)
        if pad:
            L.orc_set_initial_ues(self.h, self.U0)
        self._velnum = np.asarray([vel_number(v) for v in vel_specs[:self.U0]], dtype=np.float64)
        if (self._velnum >= 0).any():
            L.orc_set_velocity(self.h, self.U0, _p(self._velnum, _c_dp))
        if pause is not None or border is not None:                 # RandomWaypoint(pause_duration, border_buffer), movement.py:87-104
            self._pause = np.asarray([2] * self.U0 if pause is None else pause, dtype=np.int32)
            self._border = np.asarray([10] * self.U0 if border is None else border, dtype=np.int32)
            L.orc_set_movement(self.h, self.U0, _p(self._pause, _c_ip), _p(self._border, _c_ip))

    def probe_data_rate(self, b, u, ewma=None):
        """Basestation.data_rate(ue) (station.py:204-220) in the current state; ewma: replace the UEs' EWMA rates first."""
        e = None if ewma is None else np.ascontiguousarray(ewma, dtype=np.float64)
        return float(lib().orc_probe_data_rate(self.h, int(b), int(u), _p(e, _c_dp) if e is not None else None))

    def set_events(self, remove_idx=(), add_xy=()):
        """Arrival / departure of the NEXT step (base.py:433-443).  Lists: tape mode (reference draws)."""
        r = np.ascontiguousarray(remove_idx, dtype=np.int32).reshape(-1)
        a = np.ascontiguousarray(add_xy, dtype=np.int32).reshape(-1, 2)
        lib().orc_set_events(self.h, len(r), _p(r, _c_ip) if len(r) else None, len(a), _p(a, _c_ip) if len(a) else None)

    def set_event_counts(self, n_remove, n_add):
        """Philox mode: only the counts; indices / border points come from the keyed draws."""
        lib().orc_set_events(self.h, int(n_remove), None, int(n_add), None)

    def set_tape_ids(self, pos0, triples):
        pos0 = np.ascontiguousarray(pos0, dtype=np.int32)
        triples = np.ascontiguousarray(triples, dtype=np.int32)
        lib().orc_set_tape_ids(self.h, triples.shape[1], triples.shape[0], _p(pos0, _c_ip), _p(triples, _c_ip))

    def num_ue(self):
        return lib().orc_num_ue(self.h)

    def uids(self):
        u = np.zeros(self.U, dtype=np.int32)
        lib().orc_get_uids(self.h, _p(u, _c_ip))
        return u

    def end_of_episode_list(self):
        """[(id, born)] in ue_list order -- what MobileEnv.seed iterates at the next reset (base.py:138-143)."""
        ids = self.uids()
        return [(int(ids[s]), bool(lib().orc_slot_born(self.h, s))) for s in range(self.num_ue())]

    def orig_consumed(self):
        c = np.zeros(self.U0, dtype=np.int32)
        lib().orc_get_orig_consumed(self.h, _p(c, _c_ip))
        return c

    def __del__(self):
        if getattr(self, 'h', None) is not None and _lib is not None:
            _lib.orc_destroy(self.h)
            self.h = None

    def set_tape(self, pos0, triples):
        pos0 = np.ascontiguousarray(pos0, dtype=np.int32)
        triples = np.ascontiguousarray(triples, dtype=np.int32)
        lib().orc_set_tape(self.h, triples.shape[1], _p(pos0, _c_ip), _p(triples, _c_ip))

    def set_philox(self, seed, global_env_id):
        lib().orc_set_philox(self.h, seed, global_env_id)

    def set_episode(self, ep):
        lib().orc_set_episode(self.h, ep)

    def reset(self):
        lib().orc_reset(self.h)

    def step(self, action):
        a = np.ascontiguousarray(action, dtype=np.int32)
        rc = lib().orc_step(self.h, _p(a, _c_ip))
        if rc != 0:
            raise AssertionError(f"Action {action} does not fit action space")   # central.py:61
        return rc

    def obs(self):
        U, B = self.U, self.B
        out = {k: np.zeros((U, B)) for k in ('connected', 'dr', 'ues_at_bs', 'util_at_bs')}
        out['utility'] = np.zeros(U)
        lib().orc_get_obs(self.h, _p(out['connected'], _c_dp), _p(out['dr'], _c_dp), _p(out['utility'], _c_dp),
                          _p(out['ues_at_bs'], _c_dp), _p(out['util_at_bs'], _c_dp))
        return out

    def reward(self):
        r = np.zeros(1 if self.kind == CENTRAL else self.U)
        lib().orc_get_reward(self.h, _p(r, _c_dp))
        return r

    def reward_before(self):
        r = np.zeros(self.U)
        lib().orc_get_reward_before(self.h, _p(r, _c_dp))
        return r

    def state(self):
        U, B = self.U, self.B
        s = dict(pos=np.zeros((U, 2)), wp=np.zeros((U, 2)), vel=np.zeros(U), pausing=np.zeros(U, np.int32),
                 curr_pause=np.zeros(U, np.int32), conn=np.zeros((U, B), np.uint8), dr=np.zeros((U, B)),
                 curr_dr=np.zeros(U), ewma=np.zeros(U), utility=np.zeros(U), conn_order=np.zeros((B, U), np.int32))
        lib().orc_get_state(self.h, _p(s['pos'], _c_dp), _p(s['wp'], _c_dp), _p(s['vel'], _c_dp),
                            _p(s['pausing'], _c_ip), _p(s['curr_pause'], _c_ip), _p(s['conn'], _c_u8p),
                            _p(s['dr'], _c_dp), _p(s['curr_dr'], _c_dp), _p(s['ewma'], _c_dp), _p(s['utility'], _c_dp),
                            _p(s['conn_order'], _c_ip))
        return s

    def sum_utility(self):
        return lib().orc_sum_utility(self.h)

    def time(self):
        return lib().orc_time(self.h)

    def cursors(self):
        return np.array([lib().orc_tape_cursor(self.h, u) for u in range(self.U)], dtype=np.int32)


def arrival_schedule(episode_length, ue_arrival=None, new_ue_interval=None):
    """Per-step (n_remove, n_add) as MobileEnv.step applies them (base.py:433-443); the step index is env.time
    BEFORE the step.  ue_arrival disables the interval (base.py:52-56)."""
    sched = [(0, 0)] * episode_length
    if ue_arrival:
        for t, n in ue_arrival.items():
            if 0 <= int(t) < episode_length:
                sched[int(t)] = (-n, 0) if n < 0 else (0, n)
    elif new_ue_interval:
        for t in range(1, episode_length):
            if t % new_ue_interval == 0:
                sched[t] = (0, 1)
    return sched


class DynRefStreams:
    """Streams of the INITIAL UEs when UEs arrive / depart (base.py:433-443).  Their ``random.Random`` objects live as
    long as the env: at reset the reference restores the original ue_list but re-seeds (rand_episodes=False) the UEs
    of the list as it stood at the END of the previous episode, BY POSITION in that list (base.py:171-173 -> 138-143)
    -- so an initial UE that moved up in the list gets another UE's seed, and one that had left keeps its old stream.
    This class keeps that exact bookkeeping; arrived UEs are always freshly seeded (base.py:602-604)."""

    def __init__(self, seed, map_w, map_h, vel_specs, depth=48, rand_episodes=False, init_xy=None):
        self.seed, self.w, self.h, self.depth, self.rand_episodes = seed, int(map_w), int(map_h), depth, rand_episodes
        self.vel = [vel_range(v) for v in vel_specs]
        self.U0 = len(self.vel)
        self.init_xy = init_xy if init_xy is not None else [(-1, -1)] * self.U0
        self.pos_rng = [random.Random(seed + 100 * (i + 1)) for i in range(self.U0)]
        self.mov_rng = [random.Random(seed + 100 * (i + 1)) for i in range(self.U0)]
        self._states = None
        self._shift = [0] * self.U0

    def reseed_live(self, seed, trip_all, slots):
        """MobileEnv.seed(seed) while an episode with a changing UE list runs (base.py:132-143): every UE of the CURRENT list --
        initial or arrived -- gets seed + 100*(position + 1) for both streams.  slots = [(uid, born, cursor)] in list order;
        trip_all = the episode's triples, rows by id (initial UEs, then one row per id of an arriving UE).  Returns trip_all with
        the rows of the listed UEs continued from their cursor by the new movement streams.  UEs that arrive later are still
        seeded with the CONFIGURED seed (base.py:601-604), initial UEs that have left keep their old stream."""
        trip = trip_all.copy()
        depth = trip.shape[1]
        for pos, (uid, born, cur) in enumerate(slots):
            s, cur = seed + 100 * (pos + 1), int(cur)
            if born:                                             # a User created by add_new_ue: 'slow', default border (base.py:597-599)
                r, (lo, hi), row = random.Random(s), (1, 3), self.U0 + uid - 1
            else:
                self.pos_rng[uid - 1].seed(s)
                self.mov_rng[uid - 1].seed(s)
                r, (lo, hi), row = self.mov_rng[uid - 1], self.vel[uid - 1], uid - 1
                self._states[uid - 1] = [r.getstate()]
                self._shift[uid - 1] = cur
            for k in range(cur, depth):
                v = r.randint(lo, hi) if lo != hi else lo
                trip[row, k] = (v, r.randint(10, self.w - 10), r.randint(10, self.h - 10))
                if not born:
                    self._states[uid - 1].append(r.getstate())
        return trip

    def draw_episode(self, end_list=None, consumed=None):
        if self._states is not None:
            for i in range(self.U0):
                self.mov_rng[i].setstate(self._states[i][int(consumed[i]) - self._shift[i]])
        self._shift = [0] * self.U0
        if not self.rand_episodes:
            order = [(i + 1, False) for i in range(self.U0)] if end_list is None else end_list
            for pos, (uid, born) in enumerate(order):
                if not born:
                    self.pos_rng[uid - 1].seed(self.seed + 100 * (pos + 1))
                    self.mov_rng[uid - 1].seed(self.seed + 100 * (pos + 1))
        pos0 = np.zeros((self.U0, 2), dtype=np.int32)
        trip = np.zeros((self.U0, self.depth, 3), dtype=np.int32)
        self._states = []
        for i in range(self.U0):
            ix, iy = self.init_xy[i]
            pos0[i, 0] = self.pos_rng[i].randint(0, self.w) if ix < 0 else ix
            pos0[i, 1] = self.pos_rng[i].randint(0, self.h) if iy < 0 else iy
            lo, hi = self.vel[i]
            st = [self.mov_rng[i].getstate()]
            for k in range(self.depth):
                v = self.mov_rng[i].randint(lo, hi) if lo != hi else lo
                trip[i, k] = (v, self.mov_rng[i].randint(10, self.w - 10), self.mov_rng[i].randint(10, self.h - 10))
                st.append(self.mov_rng[i].getstate())
            self._states.append(st)
        return pos0, trip


class RefEventDraws:
    """The reference's draws for arrivals / departures: ``map.rng`` (map.py:52-65) and the GLOBAL ``random``
    (base.py:611), both seeded with the env seed by MobileEnv.seed (base.py:132-136)."""

    def __init__(self, seed, map_w, map_h, rand_episodes=False):
        self.seed, self.w, self.h, self.rand_episodes = seed, int(map_w), int(map_h), rand_episodes
        self._seed()

    def _seed(self):
        self.map_rng, self.glob = random.Random(self.seed), random.Random(self.seed)

    def new_episode(self):
        if not self.rand_episodes:
            self._seed()

    def reseed_live(self, seed):
        """MobileEnv.seed(seed) on a live env: the global generator and the map's restart from `seed` (base.py:134-136); the
        configured seed stays what reset() of a rand_episodes=False env goes back to."""
        self.map_rng, self.glob = random.Random(seed), random.Random(seed)

    def departures(self, n_remove, num_ue):
        out = []
        for _ in range(n_remove):
            out.append(self.glob.randint(0, num_ue - 1))
            num_ue -= 1
        return out

    def arrivals(self, n_add):
        out = []
        for _ in range(n_add):
            x, y = self.map_rng.randint(0, self.w), self.map_rng.randint(0, self.h)
            border = self.map_rng.choice(['left', 'right', 'top', 'bottom'])
            out.append({'left': (0, y), 'right': (self.w - 1, y), 'top': (x, self.h - 1), 'bottom': (x, 0)}[border])
        return out


class OracleBatch:
    """E oracle envs stepped together (OpenMP over envs): large-E parity checks and the CPU baseline."""

    def __init__(self, envs, num_threads=None):
        self.envs = list(envs)
        self.E = len(self.envs)
        self.U, self.B, self.kind = envs[0].U, envs[0].B, envs[0].kind
        self._handles = (ctypes.c_void_p * self.E)(*[e.h for e in self.envs])
        self.num_threads = num_threads or lib().orc_max_threads()
        self.obs_dim = 4 * self.B + 1 if self.kind == MULTI else 2 * self.B + 1

    def reset(self):
        obs = np.zeros((self.E, self.U, self.obs_dim), dtype=np.float32)
        lib().orc_batch_reset(self._handles, self.E, _p(obs, _c_fp), self.num_threads)
        return obs

    def step(self, action, want_obs=True):
        a = np.ascontiguousarray(action, dtype=np.uint8).reshape(self.E, self.U)
        obs = np.zeros((self.E, self.U, self.obs_dim), dtype=np.float32) if want_obs else None
        rew = np.zeros((self.E, self.U) if self.kind == MULTI else (self.E,), dtype=np.float32)
        conn = np.zeros((self.E, self.U), dtype=np.uint32)
        pos = np.zeros((self.E, self.U, 2), dtype=np.float64)
        lib().orc_batch_step(self._handles, self.E, _p(a, _c_u8p), _p(obs, _c_fp), _p(rew, _c_fp), _p(conn, _c_u32p),
                             _p(pos, _c_dp), self.num_threads)
        if self.B > 32:                          # stations 32-63: a second word per UE, as the device's state.conn_hi; returned as ONE uint64 mask
            hi = np.zeros((self.E, self.U), dtype=np.uint32)
            lib().orc_batch_conn_hi(self._handles, self.E, _p(hi, _c_u32p))
            conn = conn.astype(np.uint64) | (hi.astype(np.uint64) << np.uint64(32))
        return obs, rew, conn, pos


    def rates(self, want_dr_rel=False):
        """FP64 per-UE values of every env: curr_dr (sum of the connections' shared rates), ewma, utility [E, U] and, on
        request, the relative SNR rows dr_rel [E, U, B] before their float32 cast -- what the at-scale parity tests hold the
        device's ue_dr / ewma / obs.dr to at 1e-5 RELATIVE."""
        E, U, B = self.E, self.U, self.B
        out = {'curr_dr': np.zeros((E, U)), 'ewma': np.zeros((E, U)), 'utility': np.zeros((E, U))}
        rel = np.zeros((E, U, B)) if want_dr_rel else None
        lib().orc_batch_rates(self._handles, E, _p(out['curr_dr'], _c_dp), _p(out['ewma'], _c_dp), _p(out['utility'], _c_dp),
                              _p(rel, _c_dp), self.num_threads)
        if want_dr_rel:
            out['dr_rel'] = rel
        return out


def snr(d):
    return lib().orc_snr(float(d))


def can_connect(d):
    return bool(lib().orc_can_connect(float(d)))


def dr_unshared(d):
    return lib().orc_dr_unshared(float(d))


def log_utility(dr):
    return lib().orc_log_utility(float(dr))


def step_utility(dr, req):
    return lib().orc_step_utility(float(dr), float(req))


def connect_threshold_distance():
    return lib().orc_connect_threshold_distance()


def philox4x32_10(ctr, key):
    c = (ctypes.c_uint32 * 4)(*ctr)
    k = (ctypes.c_uint32 * 2)(*key)
    o = (ctypes.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return list(o)
