"""Config holders with the reference's constructor signatures: Map, Basestation, User, RandomWaypoint.

In the reference these objects *are* the simulation (deepcomp/env/entities/{map,station,user}.py,
env/util/movement.py).  Here the simulation state lives in device tensors; these classes only carry the
parameters an ``env_config`` dict hands to the env constructor (env_setup.py:247-256) and expose read-only
views of the device state for callers that poke at ``env.ue_list[i].pos`` and friends.  The env constructors
also accept the reference's own objects (duck-typed: ``.pos.x``, ``.sharing_model``, ``.movement.init_velocity`` ...).
"""
import math
from types import SimpleNamespace

from . import scenarios

SUPPORTED_SHARING = {'max-cap', 'resource-fair', 'rate-fair', 'proportional-fair'}   # constants.py:22
SUPPORTED_UTILITIES = {'log', 'step', 'linear'}                                     # constants.py:25


class Map:
    """map.py:12-30: rectangular playground; width/height are int()-truncated."""

    def __init__(self, width, height, min_x=0, min_y=0):
        self.width = int(width)
        self.height = int(height)
        self.min_x, self.min_y = min_x, min_y
        self.max_x, self.max_y = min_x + self.width, min_y + self.height
        self.diagonal = math.sqrt(self.width ** 2 + self.height ** 2)

    def __repr__(self):
        return f'{self.width}x{self.height}map'


class Point(SimpleNamespace):
    """Minimal (x, y) holder standing in for shapely's Point in configs."""

    def __init__(self, x, y):
        super().__init__(x=float(x), y=float(y))

    def __str__(self):
        return f"POINT ({self.x:g} {self.y:g})"


class RandomWaypoint:
    """movement.py:86-104: velocity is a number or 'slow' (1..3) / 'fast' (5..10)."""

    def __init__(self, map, velocity, pause_duration=2, border_buffer=10):
        assert border_buffer > 0, "Border Buffer must be >0 to avoid placing waypoints on or outside map borders."   # movement.py:103
        self.map = map
        self.init_velocity = velocity
        self.pause_duration = pause_duration
        self.border_buffer = border_buffer

    def __str__(self):
        return f"RandomWaypoint({self.init_velocity})"


class Basestation:
    """station.py:14-30: id, position, sharing model (channel constants are fixed in the reference)."""

    def __init__(self, id, pos, sharing_model):
        self.id = id
        self.pos = pos if hasattr(pos, 'x') else Point(*pos)
        assert sharing_model in SUPPORTED_SHARING, f"{sharing_model=} not supported. {SUPPORTED_SHARING=}"   # station.py:22
        self.sharing_model = sharing_model
        self.bw, self.frequency, self.noise, self.tx_power, self.height = 9e6, 2500, 1e-9, 30, 50
        self._env, self._idx = None, None

    def __repr__(self):
        return str(self.id)

    @property
    def num_conn_ues(self):
        return self._env._host_view()['num_conn'][self._idx] if self._env is not None else 0


class User:
    """user.py:16-46."""

    def __init__(self, id, map, pos_x, pos_y, movement, util_func='log', dr_req=1):
        self.id = id
        self.map = map
        self.movement = movement
        assert util_func in SUPPORTED_UTILITIES, f"Utility function {util_func} not supported. Supported: {SUPPORTED_UTILITIES}"
        self.util_func = util_func
        self.dr_req = dr_req
        self.init_pos_x, self.init_pos_y = pos_x, pos_y
        self._env, self._idx = None, None

    def __repr__(self):
        return str(self.id)

    def __eq__(self, other):
        return type(other) is type(self) and self.id == other.id

    def __hash__(self):
        return hash(self.id)

    # read-only views of the device state of env 0 (compat mode)
    def _v(self, key):
        if self._env is None:
            return None
        return self._env._host_view()[key][self._idx]

    @property
    def pos(self):
        p = self._v('pos')
        return None if p is None else Point(p[0], p[1])

    @property
    def curr_dr(self):
        return self._v('curr_dr')

    @property
    def utility(self):
        return self._v('utility')

    @property
    def ewma_dr(self):
        return self._v('ewma')


def build_from_scenario(scn: scenarios.Scenario):
    """Scenario table -> (Map, [Basestation], [User]) exactly as env_setup.get_env would hand to env_config."""
    m = Map(scn.width, scn.height)
    bs_list = [Basestation(i, Point(x, y), s) for i, (x, y), s in zip(scn.bs_ids, scn.bs_pos, scn.bs_sharing)]
    ue_list = [User(s['id'], m, s['pos_x'], s['pos_y'],
                    RandomWaypoint(m, velocity=s['velocity'], pause_duration=s.get('pause_duration', 2),
                                   border_buffer=s.get('border_buffer', 10)),
                    util_func=s['util_func'], dr_req=s['dr_req']) for s in scn.ue_specs]
    return m, bs_list, ue_list


def make_env_config(scn, seed=42, episode_length=100, reward='avg', rand_episodes=False, num_envs=1, ue_arrival=None,
                    new_ue_interval=None, **extra):
    """The env_config dict of env_setup.create_env_config (env_setup.py:247-256) for a scenario table."""
    m, bs_list, ue_list = build_from_scenario(scn)
    cfg = {'episode_length': episode_length, 'seed': seed, 'map': m, 'bs_list': bs_list, 'ue_list': ue_list,
           'rand_episodes': rand_episodes, 'new_ue_interval': new_ue_interval, 'reward': reward, 'max_ues': None,
           'ue_arrival': ue_arrival, 'log_metrics': True, 'dashboard': False, 'ue_details': False,
           'num_envs': num_envs}
    cfg.update(extra)
    return cfg
