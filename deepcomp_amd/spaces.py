"""Observation / action space descriptors.  Uses gym's own classes when gym is importable (the RLlib side
then sees exactly what the reference env advertises); otherwise minimal stand-ins with the attributes the
reference's callers read (``.contains``, ``.shape``, ``.n``, ``.nvec``, ``.spaces``; env_setup.py:291-309).
"""
from collections import OrderedDict

import numpy as np

try:                                            # pragma: no cover - gym is absent in the build image
    from gym.spaces import Box, Dict, Discrete, MultiBinary, MultiDiscrete   # noqa: F401
except Exception:                               # noqa: BLE001
    class _Space:
        shape = None

        def seed(self, seed=None):
            return [seed]

    class Discrete(_Space):
        def __init__(self, n):
            self.n = int(n)
            self.shape = ()

        def contains(self, x):
            try:
                xi = int(x)
            except (TypeError, ValueError):
                return False
            return xi == x and 0 <= xi < self.n

        def __repr__(self):
            return f"Discrete({self.n})"

    class MultiBinary(_Space):
        def __init__(self, n):
            self.n = int(n)
            self.shape = (self.n,)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(((x == 0) | (x == 1)).all())

        def __repr__(self):
            return f"MultiBinary({self.n})"

    class MultiDiscrete(_Space):
        def __init__(self, nvec):
            self.nvec = np.asarray(nvec, dtype=np.int64)
            self.shape = self.nvec.shape

        def contains(self, x):
            try:
                x = np.asarray(x, dtype=np.int64)
            except (TypeError, ValueError):
                return False
            return x.shape == self.shape and bool((0 <= x).all() and (x < self.nvec).all())

        def __repr__(self):
            return f"MultiDiscrete({self.nvec.tolist()})"

    class Box(_Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool((x >= self.low).all() and (x <= self.high).all())

        def __repr__(self):
            return f"Box({self.low}, {self.high}, {self.shape})"

    class Dict(_Space):
        def __init__(self, spaces):
            if isinstance(spaces, dict) and not isinstance(spaces, OrderedDict):
                spaces = OrderedDict(sorted(spaces.items()))     # gym<0.23 sorts plain dicts by key
            self.spaces = spaces

        def __repr__(self):
            return "Dict(" + ", ".join(f"{k}:{v}" for k, v in self.spaces.items()) + ")"
