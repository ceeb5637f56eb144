"""Host-side draw tapes for the reference-exact RNG mode.

The reference gives every UE two ``random.Random`` streams seeded ``seed + 100*(i+1)`` (base.py:132-143,
user.py:94-96) and draws, per reset, the start position (user.py:98-109) and, per movement reset, an optional
velocity plus a waypoint (movement.py:110-130).  The device kernels consume those draws from a pre-drawn
tape.  Two producers of the same tape:

* ``mt_tape``  -- the C++ CPython-compatible MT19937 inside libdcomp_hip.so (fast, stateless: always the
  first ``depth`` triples of freshly seeded streams; this is all ``rand_episodes=False`` needs because the
  reference re-seeds at every reset, base.py:171-173);
* ``StdlibStreams`` -- the stdlib generator itself, kept alive across episodes for ``rand_episodes=True``
  (streams continue where the previous episode stopped).
"""
import ctypes
import random

import numpy as np

from . import _lib


def vel_range(v):
    """Velocity spec -> inclusive draw range (movement.py:112-117)."""
    if v == 'slow':
        return 1, 3
    if v == 'fast':
        return 5, 10
    return int(v), int(v)


def mt_tape(cfg_struct, seeds, depth):
    """(pos0[E*U,2] int32, triples[E*U,depth,4] uint16) from the library's MT19937."""
    L = _lib.load()
    seeds = np.ascontiguousarray(seeds, dtype=np.int64)
    E, U = len(seeds), cfg_struct.num_ue
    pos0 = np.zeros((E * U, 2), dtype=np.int32)
    trip = np.zeros((E * U, depth, 4), dtype=np.uint16)
    _lib.check(L.dcomp_mt_draw_tape(ctypes.byref(cfg_struct), seeds.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), E, depth,
                                    pos0.ctypes.data, trip.ctypes.data))
    return pos0, trip


class StdlibStreams:
    """Per-UE stdlib ``random.Random`` pairs for E envs; supports continuing streams across episodes."""

    def __init__(self, seeds, map_w, map_h, vel_specs, init_xy, depth, border=None):
        self.border = [10] * len(vel_specs) if border is None else [int(b) for b in border]      # movement.py:87,126-127
        self.seeds = [int(s) for s in seeds]
        self.w, self.h, self.depth = int(map_w), int(map_h), int(depth)
        self.vel = [vel_range(v) for v in vel_specs]
        self.init_xy = list(init_xy)
        self.U = len(self.vel)
        self._seed_all()
        self._states = None
        self._shift = None
        self._trip = None

    def _seed_all(self):
        self.pos_rng = [[random.Random(s + 100 * (i + 1)) for i in range(self.U)] for s in self.seeds]
        self.mov_rng = [[random.Random(s + 100 * (i + 1)) for i in range(self.U)] for s in self.seeds]

    def reseed_live(self, seeds, consumed=None):
        """MobileEnv.seed() on a live env (base.py:132-143): BOTH streams of every UE start over at once, from the new seeds.
        What a UE still draws in the running episode comes from the start of its new movement stream: returns the episode's
        triples with row (e, i) continued from position consumed[e*U+i] (the cursor field of the state) by the new stream --
        None when no episode has been drawn yet (the next reset() then simply starts the new streams)."""
        self.seeds = [int(s) for s in seeds]
        self._seed_all()
        if self._trip is None:
            self._states = None
            return None
        E, U, D = len(self.seeds), self.U, self.depth
        trip = self._trip.copy()
        self._states, self._shift = [], [int(c) for c in consumed]
        for e in range(E):
            st_e = []
            for i in range(U):
                mr = self.mov_rng[e][i]
                lo, hi = self.vel[i]
                st = [mr.getstate()]
                for k in range(self._shift[e * U + i], D):
                    trip[e * U + i, k, 0] = mr.randint(lo, hi) if lo != hi else lo
                    trip[e * U + i, k, 1] = mr.randint(self.border[i], self.w - self.border[i])
                    trip[e * U + i, k, 2] = mr.randint(self.border[i], self.h - self.border[i])
                    st.append(mr.getstate())
                st_e.append(st)
            self._states.append(st_e)
        self._trip = trip
        return self._pos0, trip

    def draw_episode(self, reseed, consumed=None):
        """consumed[E*U]: movement triples the previous episode used (the cursor field of the state)."""
        E, U, D = len(self.seeds), self.U, self.depth
        if reseed:
            self._seed_all()
        elif self._states is not None:
            assert consumed is not None
            for e in range(E):
                for i in range(U):
                    shift = self._shift[e * U + i] if self._shift else 0      # triples drawn before a live re-seed
                    self.mov_rng[e][i].setstate(self._states[e][i][int(consumed[e * U + i]) - shift])
        self._shift = None
        pos0 = np.zeros((E * U, 2), dtype=np.int32)
        trip = np.zeros((E * U, D, 4), dtype=np.uint16)
        self._states = []
        for e in range(E):
            st_e = []
            for i in range(U):
                ix, iy = self.init_xy[i]
                pr, mr = self.pos_rng[e][i], self.mov_rng[e][i]
                pos0[e * U + i, 0] = pr.randint(0, self.w) if ix < 0 else ix
                pos0[e * U + i, 1] = pr.randint(0, self.h) if iy < 0 else iy
                lo, hi = self.vel[i]
                st = [mr.getstate()]
                for k in range(D):
                    v = mr.randint(lo, hi) if lo != hi else lo
                    trip[e * U + i, k, 0] = v
                    trip[e * U + i, k, 1] = mr.randint(self.border[i], self.w - self.border[i])
                    trip[e * U + i, k, 2] = mr.randint(self.border[i], self.h - self.border[i])
                    st.append(mr.getstate())
                st_e.append(st)
            self._states.append(st_e)
        self._pos0, self._trip = pos0, trip
        return pos0, trip

    def extend(self, new_depth):
        """Continue every movement stream beyond the tape drawn by draw_episode (an episode that runs longer than the
        tape was sized for: --cont-train never resets, main.py:48-51).  The generators still stand where the tape ended."""
        E, U, D = len(self.seeds), self.U, self.depth
        trip = np.zeros((E * U, new_depth, 4), dtype=np.uint16)
        trip[:, :D] = self._trip
        for e in range(E):
            for i in range(U):
                mr, st = self.mov_rng[e][i], self._states[e][i]
                lo, hi = self.vel[i]
                for k in range(D, new_depth):
                    trip[e * U + i, k, 0] = mr.randint(lo, hi) if lo != hi else lo
                    trip[e * U + i, k, 1] = mr.randint(self.border[i], self.w - self.border[i])
                    trip[e * U + i, k, 2] = mr.randint(self.border[i], self.h - self.border[i])
                    st.append(mr.getstate())
        self.depth, self._trip = new_depth, trip
        return self._pos0, trip


def arrival_schedule(episode_length, ue_arrival=None, new_ue_interval=None):
    """Per-step (n_remove, n_add) as MobileEnv.step applies them (base.py:433-443), indexed by env.time before the
    step.  A ue_arrival sequence disables the interval (base.py:52-56)."""
    sched = [(0, 0)] * int(episode_length)
    if ue_arrival:
        for t, n in ue_arrival.items():
            t, n = int(t), int(n)
            if 0 <= t < episode_length:
                sched[t] = (-n, 0) if n < 0 else (0, n)
    elif new_ue_interval:
        for t in range(1, int(episode_length)):
            if t % int(new_ue_interval) == 0:
                sched[t] = (0, 1)
    return sched


def max_num_ue(num_ue, episode_length, ue_arrival=None, new_ue_interval=None):
    """MobileEnv.get_max_num_ue (base.py:191-210)."""
    max_ues = num_ue
    if new_ue_interval is not None and not ue_arrival:
        max_ues = num_ue + int((episode_length - 1) / new_ue_interval)
    if ue_arrival:
        cur = max_ues = num_ue
        for _, n in sorted((int(t), int(n)) for t, n in ue_arrival.items()):
            cur += n
            max_ues = max(max_ues, cur)
    return max_ues


class DynamicStdlibStreams:
    """Reference-exact draws when UEs arrive / depart (`rng='reference'`), for E envs.

    * initial UEs: ``random.Random`` pairs that live as long as the env.  At reset the reference re-seeds
      (rand_episodes=False) the UEs of the list as it stood at the END of the previous episode, by their POSITION in
      that list (base.py:171-173 -> 138-143), then restores the original list (base.py:177-182): an initial UE that
      moved up gets another seed, one that had left keeps its old stream.  Kept exactly.
    * arriving UEs: always freshly seeded ``seed + 100*id`` and 'slow' (base.py:592-606): one tape per possible id.
    * departures: ``random.randint(0, num_ue-1)`` of the GLOBAL generator (base.py:611); arrivals:
      ``map.rand_border_point()`` (map.py:52-65); both seeded with the env seed by MobileEnv.seed (base.py:132-136).
    """

    def __init__(self, seeds, map_w, map_h, vel_specs, init_xy, depth, rand_episodes, max_id, border=None):
        self.border = [10] * len(vel_specs) if border is None else [int(b) for b in border]      # initial UEs; arriving ones: 10
        self.seeds = [int(s) for s in seeds]
        self.w, self.h, self.depth, self.rand_episodes = int(map_w), int(map_h), int(depth), bool(rand_episodes)
        self.vel = [vel_range(v) for v in vel_specs]
        self.init_xy = list(init_xy)
        self.U0, self.max_id = len(self.vel), int(max_id)
        self.pos_rng = [[random.Random(s + 100 * (i + 1)) for i in range(self.U0)] for s in self.seeds]
        self.mov_rng = [[random.Random(s + 100 * (i + 1)) for i in range(self.U0)] for s in self.seeds]
        self.map_rng = [random.Random(s) for s in self.seeds]
        self.glob = [random.Random(s) for s in self.seeds]
        self._states = None
        self._shift = None          # [e][i]: triples initial UE i had drawn before a live re-seed (reseed_live)
        self._live_born = {}        # (e, id) -> (seed, cursor): arrived UEs re-seeded live, for extend()

    def _born_row(self, trip, row, seed, cursor, upto):
        """Row of an arrived UE re-seeded live: 'slow', default border (base.py:597-599), new stream from `cursor` on."""
        r = random.Random(seed)
        for k in range(cursor, upto):
            trip[row, k, :3] = (r.randint(1, 3), r.randint(10, self.w - 10), r.randint(10, self.h - 10))

    def reseed_live(self, seeds, lists, cursors):
        """MobileEnv.seed() while an episode with a changing UE list runs (base.py:132-143).  Per env: the global generator
        (departures, base.py:611) and the map's (arrival points, map.py:52-65) restart from the new seed; every UE of the CURRENT
        list -- initial or arrived -- gets `seed + 100*(position + 1)` for both streams.  lists[e] = [(id, born)] in list order,
        cursors[e][slot] = movement triples consumed so far.  Returns (pos0, triples) with the rows of the listed UEs continued
        from their cursor by the new streams.  UEs that arrive later are still seeded with the CONFIGURED seed (base.py:601-604:
        env_seed), initial UEs that have left keep their old stream, and reset() of a rand_episodes=False env re-seeds with the
        configured seed as always (self.seeds is not touched)."""
        E, U0, D = len(self.seeds), self.U0, self.depth
        ids = U0 + self.max_id
        trip = self._trip.copy()
        if self._shift is None:
            self._shift = [[0] * U0 for _ in range(E)]
        for e in range(E):
            s = int(seeds[e])
            self.map_rng[e].seed(s)
            self.glob[e].seed(s)
            for pos, (uid, born) in enumerate(lists[e]):
                sd, cur = s + 100 * (pos + 1), int(cursors[e][pos])
                if born:
                    self._live_born[(e, uid)] = (sd, cur)
                    self._born_row(trip, e * ids + U0 + uid - 1, sd, cur, D)
                    continue
                i = uid - 1
                self.pos_rng[e][i].seed(sd)
                mr = self.mov_rng[e][i]
                mr.seed(sd)
                lo, hi = self.vel[i]
                st = [mr.getstate()]
                for k in range(cur, D):
                    trip[e * ids + i, k, :3] = (mr.randint(lo, hi) if lo != hi else lo, mr.randint(self.border[i], self.w - self.border[i]),
                                                mr.randint(self.border[i], self.h - self.border[i]))
                    st.append(mr.getstate())
                self._states[e][i], self._shift[e][i] = st, cur
        self._trip = trip
        return self._pos0, trip

    def draw_episode(self, end_lists=None, consumed=None):
        """end_lists[e] = [(id, born)] of env e's list at the end of the previous episode; consumed[e][i] = movement
        triples initial UE i used in it.  Returns (pos0[E*U0,2], triples[E*(U0+max_id), depth, 4])."""
        E, U0, D = len(self.seeds), self.U0, self.depth
        if self._states is not None:
            for e in range(E):
                for i in range(U0):
                    shift = self._shift[e][i] if self._shift else 0
                    self.mov_rng[e][i].setstate(self._states[e][i][int(consumed[e][i]) - shift])
        self._shift, self._live_born = None, {}
        if not self.rand_episodes:
            for e, s in enumerate(self.seeds):
                self.map_rng[e].seed(s)
                self.glob[e].seed(s)
                order = [(i + 1, False) for i in range(U0)] if end_lists is None else end_lists[e]
                for pos, (uid, born) in enumerate(order):
                    if not born:
                        self.pos_rng[e][uid - 1].seed(s + 100 * (pos + 1))
                        self.mov_rng[e][uid - 1].seed(s + 100 * (pos + 1))
        ids = U0 + self.max_id
        pos0 = np.zeros((E * U0, 2), dtype=np.int32)
        trip = np.zeros((E * ids, D, 4), dtype=np.uint16)
        self._states = []
        for e, s in enumerate(self.seeds):
            st_e = []
            for i in range(U0):
                ix, iy = self.init_xy[i]
                pr, mr = self.pos_rng[e][i], self.mov_rng[e][i]
                pos0[e * U0 + i] = (pr.randint(0, self.w) if ix < 0 else ix, pr.randint(0, self.h) if iy < 0 else iy)
                lo, hi = self.vel[i]
                st = [mr.getstate()]
                for k in range(D):
                    trip[e * ids + i, k, :3] = (mr.randint(lo, hi) if lo != hi else lo, mr.randint(self.border[i], self.w - self.border[i]),
                                                mr.randint(self.border[i], self.h - self.border[i]))
                    st.append(mr.getstate())
                st_e.append(st)
            self._states.append(st_e)
            for j in range(self.max_id):                      # arriving UE with id j+1: fresh stream, 'slow'
                mr = random.Random(s + 100 * (j + 1))
                for k in range(D):
                    trip[e * ids + U0 + j, k, :3] = (mr.randint(1, 3), mr.randint(10, self.w - 10), mr.randint(10, self.h - 10))
        self._pos0, self._trip = pos0, trip
        return pos0, trip

    def extend(self, new_depth):
        """As StdlibStreams.extend: initial UEs continue their generators, the (stateless) tapes of arriving ids are redrawn."""
        E, U0, D = len(self.seeds), self.U0, self.depth
        ids = U0 + self.max_id
        trip = np.zeros((E * ids, new_depth, 4), dtype=np.uint16)
        trip[:, :D] = self._trip
        for e, s in enumerate(self.seeds):
            for i in range(U0):
                mr, st = self.mov_rng[e][i], self._states[e][i]
                lo, hi = self.vel[i]
                for k in range(D, new_depth):
                    trip[e * ids + i, k, :3] = (mr.randint(lo, hi) if lo != hi else lo, mr.randint(self.border[i], self.w - self.border[i]),
                                                mr.randint(self.border[i], self.h - self.border[i]))
                    st.append(mr.getstate())
            for j in range(self.max_id):
                mr = random.Random(s + 100 * (j + 1))
                for k in range(new_depth):
                    trip[e * ids + U0 + j, k, :3] = (mr.randint(1, 3), mr.randint(10, self.w - 10), mr.randint(10, self.h - 10))
                if (e, j + 1) in self._live_born:               # an arrived UE that seed() re-seeded while the episode ran
                    sd, cur = self._live_born[(e, j + 1)]
                    self._born_row(trip, e * ids + U0 + j, sd, cur, new_depth)
        self.depth, self._trip = new_depth, trip
        return self._pos0, trip

    def departures(self, n_remove, num_ue):
        out = np.zeros((len(self.seeds), n_remove), dtype=np.int32)
        for e in range(len(self.seeds)):
            n = num_ue
            for k in range(n_remove):
                out[e, k] = self.glob[e].randint(0, n - 1)
                n -= 1
        return out

    def arrivals(self, n_add):
        out = np.zeros((len(self.seeds), n_add, 2), dtype=np.int32)
        for e in range(len(self.seeds)):
            r = self.map_rng[e]
            for k in range(n_add):
                x, y = r.randint(0, self.w), r.randint(0, self.h)
                border = r.choice(['left', 'right', 'top', 'bottom'])
                out[e, k] = {'left': (0, y), 'right': (self.w - 1, y), 'top': (x, self.h - 1), 'bottom': (x, 0)}[border]
        return out
