"""CPU tests of the drop-in boundary: class hierarchy under gym / ray, host-side RNG tapes.

The reference's classes are gym.Env / RLlib MultiAgentEnv subclasses (deepcomp/env/single_ue/base.py:20,
deepcomp/env/multi_ue/multi_agent.py:1-6) and its callers dispatch on `MultiAgentEnv in env_class.__mro__`
(deepcomp/util/env_setup.py:289, deepcomp/util/simulation.py:46).  gym and ray are absent from the build image, so the
hierarchy is checked in a subprocess that plants stand-in `gym` / `ray.rllib.env.multi_agent_env` modules in sys.modules
BEFORE importing deepcomp_amd -- exactly the import the real packages would satisfy."""
import os
import subprocess
import sys
import textwrap

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PROBE = textwrap.dedent('''
    import sys, types
    gym = types.ModuleType('gym')
    class Env:                                   # gym.Env
        metadata = {'render.modes': []}
    gym.Env = Env
    sys.modules['gym'] = gym
    ray = types.ModuleType('ray'); rllib = types.ModuleType('ray.rllib'); renv = types.ModuleType('ray.rllib.env')
    mae = types.ModuleType('ray.rllib.env.multi_agent_env')
    class MultiAgentEnv:                         # ray.rllib.env.multi_agent_env.MultiAgentEnv
        pass
    mae.MultiAgentEnv = MultiAgentEnv
    ray.rllib = rllib; rllib.env = renv; renv.multi_agent_env = mae
    for name, m in (('ray', ray), ('ray.rllib', rllib), ('ray.rllib.env', renv), ('ray.rllib.env.multi_agent_env', mae)):
        sys.modules[name] = m
    sys.path.insert(0, %r)
    from deepcomp_amd import env
    assert env.HAVE_GYM and env.HAVE_RAY
    # what the reference's callers test (env_setup.py:289, simulation.py:46)
    assert MultiAgentEnv in env.MultiAgentMobileEnv.__mro__
    assert MultiAgentEnv not in env.CentralRelNormEnv.__mro__ and MultiAgentEnv not in env.RelNormEnv.__mro__
    for cls in (env.CentralRelNormEnv, env.MultiAgentMobileEnv, env.RelNormEnv):
        assert issubclass(cls, Env), cls                       # base.py:20
        assert cls.__mro__.index(env._RefSurfaceEnv) < cls.__mro__.index(Env)   # the env's own reset/step win
    mro = env.MultiAgentMobileEnv.__mro__
    assert mro.index(env._RefSurfaceEnv) < mro.index(MultiAgentEnv)             # multi_agent.py:6: (RelNormEnv, MultiAgentEnv)
    for cls in (env.CentralRelNormEnv, env.MultiAgentMobileEnv, env.RelNormEnv):
        for m in ('reset', 'step', 'seed', 'get_num_diff_ues', 'get_max_num_ue', 'done', 'info'):
            assert callable(getattr(cls, m)), (cls, m)
    assert env.get_env_class('multi') is env.MultiAgentMobileEnv and env.get_env_class('central') is env.CentralRelNormEnv
    print('MRO-OK')
''')


def test_classes_derive_from_gym_and_rllib_bases_when_importable():
    r = subprocess.run([sys.executable, '-c', _PROBE % REPO], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and 'MRO-OK' in r.stdout, r.stdout


def test_classes_import_without_gym_and_ray():
    from deepcomp_amd import env
    assert not env.HAVE_GYM or 'gym' in sys.modules
    assert env._MultiAgentEnv in env.MultiAgentMobileEnv.__mro__
    assert env._MultiAgentEnv not in env.CentralRelNormEnv.__mro__


def _streams(depth, cls='plain'):
    from deepcomp_amd import rng
    seeds = [42, 20042, 40042]
    vel, xy = ['slow', 'fast', 3, 'slow'], [(-1, -1), (5, -1), (-1, 7), (-1, -1)]
    if cls == 'plain':
        return rng.StdlibStreams(seeds, 150, 110, vel, xy, depth)
    return rng.DynamicStdlibStreams(seeds, 150, 110, vel, xy, depth, True, 3)


def test_extended_tape_continues_the_streams():
    """An episode that outlives its tape (the reference never ends one: done() is None, --cont-train never resets) gets
    the SAME draws continued: extend(n) == a tape of depth n drawn in one go, and the generator states kept for the next
    episode line up."""
    for cls in ('plain', 'dyn'):
        a, b = _streams(6, cls), _streams(25, cls)
        if cls == 'plain':
            pa, ta = a.draw_episode(reseed=False); pb, tb = b.draw_episode(reseed=False)
        else:
            pa, ta = a.draw_episode(); pb, tb = b.draw_episode()
        assert np.array_equal(ta, tb[:, :6]) and np.array_equal(pa, pb)
        pa2, ta2 = a.extend(25)
        assert np.array_equal(ta2, tb) and np.array_equal(pa2, pb)
        assert a.depth == 25
        for e in range(3):
            for i in range(4):
                assert len(a._states[e][i]) == len(b._states[e][i]) == 26
                assert a._states[e][i][20] == b._states[e][i][20]
