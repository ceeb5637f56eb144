"""deepcomp_amd -- MI355X-native, vectorised implementation of DeepCoMP's cellular-CoMP environment step.

Hot path only: random-waypoint movement, Okumura-Hata SNR, connect/drop masks, resource-shared
Shannon rates, log utility, observation / reward packing -- one fused hand-written gfx950 kernel behind a C ABI
(include/dcomp.h), with the reference's gym / RLlib class surface on top (deepcomp_amd.env).
"""
__version__ = '0.1.0'

from . import scenarios  # noqa: F401
from .entities import Basestation, Map, Point, RandomWaypoint, User, make_env_config  # noqa: F401


def __getattr__(name):
    # env classes need torch + the HIP extension; import them lazily so scenario / config code stays importable
    if name in ('BatchedMobileEnv', 'CentralRelNormEnv', 'MultiAgentMobileEnv', 'RelNormEnv', 'get_env_class'):
        from . import env
        return getattr(env, name)
    raise AttributeError(name)
