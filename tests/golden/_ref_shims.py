"""Throw-away stand-ins for the six third-party packages the reference env imports but this
container lacks (shapely, gym, ray, structlog, structlog_round, svgpath2mpl).

TEST INFRASTRUCTURE ONLY.  Used by ``gen_golden.py`` (in the build container, where
``/root/reference`` exists) to import the *unmodified* reference env modules and record golden
vectors.  Nothing here is product code and nothing here is reference code: every class below is a
minimal re-statement of the *third-party* API surface the reference touches (SURVEY.md §8c lists
it).  The geometry is the textbook arithmetic GEOS uses for point/point distance and
point-in-axis-aligned-rectangle tests.
"""
import math
import sys
import types
from collections import OrderedDict

import numpy as np


# ----------------------------------------------------------------------------- shapely.geometry
class _Ring:
    def __init__(self, pts):
        self._pts = list(pts)

    @property
    def xy(self):
        xs = [p[0] for p in self._pts] + [self._pts[0][0]]
        ys = [p[1] for p in self._pts] + [self._pts[0][1]]
        return xs, ys


class Polygon:
    """Axis-aligned rectangle is all the reference ever builds (map.py:27, station.py:38)."""

    def __init__(self, pts):
        self._pts = [(float(x), float(y)) for x, y in pts]
        xs = [p[0] for p in self._pts]
        ys = [p[1] for p in self._pts]
        self.bounds = (min(xs), min(ys), max(xs), max(ys))
        self.exterior = _Ring(self._pts)


class Point:
    def __init__(self, *args):
        if len(args) == 1:
            x, y = args[0][0], args[0][1]
        else:
            x, y = args
        self.x = float(x)
        self.y = float(y)

    def distance(self, other):
        # GEOS Coordinate::distance: sqrt(dx*dx + dy*dy), no fused multiply-add
        dx = self.x - other.x
        dy = self.y - other.y
        return math.sqrt(dx * dx + dy * dy)

    def within(self, poly):
        x0, y0, x1, y1 = poly.bounds
        return x0 < self.x < x1 and y0 < self.y < y1

    def touches(self, poly):
        x0, y0, x1, y1 = poly.bounds
        inside_closed = x0 <= self.x <= x1 and y0 <= self.y <= y1
        return inside_closed and not self.within(poly)

    def buffer(self, r):
        n = 16
        return Polygon([(self.x + r * math.cos(2 * math.pi * i / n), self.y + r * math.sin(2 * math.pi * i / n))
                        for i in range(n)])

    def __eq__(self, other):
        return type(other) is type(self) and self.x == other.x and self.y == other.y

    def __hash__(self):
        return hash((self.x, self.y))

    def __str__(self):
        return f"POINT ({self.x:g} {self.y:g})"


# ----------------------------------------------------------------------------- gym
class _Space:
    shape = None

    def seed(self, seed=None):
        return [seed]


class Discrete(_Space):
    def __init__(self, n):
        self.n = n
        self.shape = ()

    def contains(self, x):
        try:
            xi = int(x)
        except (TypeError, ValueError):
            return False
        return xi == x and 0 <= xi < self.n


class MultiBinary(_Space):
    def __init__(self, n):
        self.n = n
        self.shape = (n,)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(((x == 0) | (x == 1)).all())


class MultiDiscrete(_Space):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = self.nvec.shape

    def contains(self, x):
        x = np.asarray(x, dtype=np.int64)
        return x.shape == self.shape and bool((0 <= x).all() and (x < self.nvec).all())


class Box(_Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool((x >= self.low).all() and (x <= self.high).all())


class Dict(_Space):
    def __init__(self, spaces):
        if isinstance(spaces, dict) and not isinstance(spaces, OrderedDict):
            spaces = OrderedDict(sorted(spaces.items()))   # gym<0.23 sorts plain dicts by key
        self.spaces = spaces


class Env:
    def __init__(self):
        pass


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


class _Logger:
    def __init__(self, **kw):
        pass

    def bind(self, **kw):
        return self

    def _noop(self, *a, **kw):
        return None

    info = debug = warning = error = _noop


def install():
    """Register the stand-ins in sys.modules (idempotent)."""
    if 'shapely' in sys.modules and getattr(sys.modules['shapely'], '_dcomp_shim', False):
        return
    geometry = _module('shapely.geometry', Point=Point, Polygon=Polygon)
    sys.modules['shapely'] = _module('shapely', geometry=geometry, _dcomp_shim=True)
    sys.modules['shapely.geometry'] = geometry

    spaces = _module('gym.spaces', Discrete=Discrete, MultiBinary=MultiBinary, MultiDiscrete=MultiDiscrete,
                     Box=Box, Dict=Dict)
    logger = _module('gym.logger', ERROR=40, set_level=lambda lvl: None)
    sys.modules['gym'] = _module('gym', Env=Env, spaces=spaces, logger=logger)
    sys.modules['gym.spaces'] = spaces
    sys.modules['gym.logger'] = logger

    stdlib = _module('structlog.stdlib', LoggerFactory=lambda *a, **k: None, filter_by_level=None)
    dev = _module('structlog.dev', ConsoleRenderer=lambda *a, **k: None)
    sys.modules['structlog'] = _module('structlog', get_logger=lambda *a, **kw: _Logger(),
                                       configure=lambda **kw: None, stdlib=stdlib, dev=dev)
    sys.modules['structlog.stdlib'] = stdlib
    sys.modules['structlog.dev'] = dev
    sys.modules['structlog_round'] = _module('structlog_round', FloatRounder=lambda **kw: None)

    from matplotlib.path import Path as _MplPath
    sys.modules['svgpath2mpl'] = _module(
        'svgpath2mpl', parse_path=lambda s: _MplPath(np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])))

    class MultiAgentEnv:   # marker base only (multi_agent.py:14: "doesn't have an __init__")
        pass

    mae = _module('ray.rllib.env.multi_agent_env', MultiAgentEnv=MultiAgentEnv)
    env = _module('ray.rllib.env', multi_agent_env=mae)
    rllib = _module('ray.rllib', env=env)
    sys.modules['ray'] = _module('ray', rllib=rllib)
    sys.modules['ray.rllib'] = rllib
    sys.modules['ray.rllib.env'] = env
    sys.modules['ray.rllib.env.multi_agent_env'] = mae
