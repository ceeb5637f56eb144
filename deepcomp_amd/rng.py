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

    def __init__(self, seeds, map_w, map_h, vel_specs, init_xy, depth):
        self.seeds = [int(s) for s in seeds]
        self.w, self.h, self.depth = int(map_w), int(map_h), int(depth)
        self.vel = [vel_range(v) for v in vel_specs]
        self.init_xy = list(init_xy)
        self.U = len(self.vel)
        self._seed_all()
        self._states = None

    def _seed_all(self):
        self.pos_rng = [[random.Random(s + 100 * (i + 1)) for i in range(self.U)] for s in self.seeds]
        self.mov_rng = [[random.Random(s + 100 * (i + 1)) for i in range(self.U)] for s in self.seeds]

    def draw_episode(self, reseed, consumed=None):
        """consumed[E*U]: movement triples the previous episode used (the cursor field of the state)."""
        E, U, D = len(self.seeds), self.U, self.depth
        if reseed:
            self._seed_all()
        elif self._states is not None:
            assert consumed is not None
            for e in range(E):
                for i in range(U):
                    self.mov_rng[e][i].setstate(self._states[e][i][int(consumed[e * U + i])])
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
                    trip[e * U + i, k, 1] = mr.randint(10, self.w - 10)
                    trip[e * U + i, k, 2] = mr.randint(10, self.h - 10)
                    st.append(mr.getstate())
                st_e.append(st)
            self._states.append(st_e)
        return pos0, trip
