"""RLlib-facing batched adapters.

RLlib 1.4 (the reference's pinned version, setup.py:14) steps several env copies per rollout worker through its
``VectorEnv`` protocol -- ``vector_reset() / reset_at(i) / vector_step(actions) / get_unwrapped()`` -- and multi-agent
envs through ``BaseEnv.poll() / send_actions() / try_reset()``.  ``CentralVectorEnv`` and ``MultiAgentBaseEnv`` expose
ONE ``BatchedMobileEnv`` (E envs, one kernel launch per step) through those protocols: one device->host copy of the packed
observation tensor per step, then per-env *views* (no per-env device work).  The observation dicts have the keys and
shapes of the reference's spaces (central.py:147-151, variants.py:255-269); the packed tensor already IS RLlib's
flattening of them (DictFlatteningPreprocessor concatenates the sub-spaces in gym's sorted-key order: connected, dr,
[ues_at_bs, util_at_bs,] utility) -- ``flatten_obs`` spells that order out and the tests hold the two against each other
and against the reference-run fixtures.

Caller contract kept (deepcomp/util/env_setup.py:262-316, deepcomp/util/simulation.py:143): ``observation_space``,
``action_space``, agent ids = ``ue.id`` strings, ``horizon`` = episode_length -> all envs of a batch reach it in the same step.

Two ways to consume a batch:
* the protocol methods (per-env Python dicts; what an unmodified RLlib sampler calls).  Since round 5 the per-env observation
  dicts are built ONCE, as numpy views over one persistent pinned host buffer that every step refills with one asynchronous
  device->host copy (``BatchedMobileEnv.outputs_host(persistent=True)``): ``vector_step`` / ``poll`` allocate no per-env object
  for observations, ``dones`` / multi-agent ``infos`` are shared constants resp. one dict per step.  The views alias the buffer:
  a step's observations must be consumed (RLlib's preprocessor copies them while flattening) before the next step overwrites
  them.  A RESET never touches them: reset observations live in buffers of their own (two, used alternately), so when RLlib's
  sampler resets env 0 at the horizon (``try_reset(0)`` / ``reset_at(0)``: the whole batch resets) the last step's observations of
  envs 1 .. E-1, which it has not preprocessed yet, are still what the step wrote (round 6).
  ``env_config['persistent_views'] = False`` restores fresh arrays per step.  ``env_config['info_level']``: 'full' (default,
  the reference's info dicts, base.py:383-411), 'scalar' (time + scalar_metrics) or 'none' (time only) for the central env,
  whose per-UE metric dicts are the remaining per-env Python work.  Measured rates: INTEGRATION.md section 1, tools/adapter_rate.py;
* ``env_config['flat_obs'] = True`` (round 6): ``observation_space`` IS the flattened ``Box`` -- per agent ``4B + 1`` floats, central
  ``U (2B + 1)``, in exactly the order RLlib's Dict-flattening preprocessor would produce (sorted keys: connected, dr, [ues_at_bs,
  util_at_bs,] utility) -- and ``vector_step`` / ``poll`` hand out ROWS of the one pinned array (``host[e]`` / ``host[e, i]``), no
  dicts.  RLlib then picks its no-op preprocessor for the Box, and the untouched fully connected policy of env_setup.py:273-283 sees the
  same input vector as with the Dict space (the tests hold the flat rows to ``flatten_obs`` of the reference-run dicts).  The Dict
  spaces stay the default: they are the reference's (central.py:147-151, variants.py:255-269);
* ``poll_tensors() / send_action_tensor()`` -- the same data as device tensors ``[E, U, 4B+1]`` / ``[E, U(2B+1)]`` with no
  host copy and no per-env objects: what a learner on the same GPU (or a custom sampler) uses at E = 65 536.

``ray`` is not installed in the build image: the classes derive from RLlib's base classes when importable and are plain
duck-typed classes otherwise.  All envs of a batch run in lock step (shared ``time``).
"""
import numpy as np
import torch

from . import spaces
from .env import BatchedMobileEnv

try:                                            # pragma: no cover - ray is absent in the build image
    from ray.rllib.env.vector_env import VectorEnv as _VectorEnvBase
    from ray.rllib.env.base_env import BaseEnv as _BaseEnvBase
except Exception:                               # noqa: BLE001
    class _VectorEnvBase:
        def __init__(self, observation_space, action_space, num_envs):
            self.observation_space, self.action_space, self.num_envs = observation_space, action_space, num_envs

    class _BaseEnvBase:
        pass

CENTRAL_KEYS = ('connected', 'dr', 'utility')                                   # sorted: gym.spaces.Dict order
MULTI_KEYS = ('connected', 'dr', 'ues_at_bs', 'util_at_bs', 'utility')


def flatten_obs(obs):
    """What RLlib's DictFlatteningPreprocessor makes of one observation dict: sub-spaces in sorted-key order, each raveled."""
    return np.concatenate([np.asarray(obs[k], dtype=np.float32).ravel() for k in sorted(obs)])


def _core_from_config(env_config, kind):
    return BatchedMobileEnv(env_config['map'], env_config['bs_list'], env_config['ue_list'], kind,
                            num_envs=int(env_config.get('num_envs', 1)), seed=env_config['seed'],
                            episode_length=env_config['episode_length'], reward=env_config['reward'],
                            rand_episodes=env_config['rand_episodes'], rng=env_config.get('rng', 'philox'),
                            device=env_config.get('device', 'cuda'), env_id_base=env_config.get('env_id_base', 0),
                            env_seeds=env_config.get('env_seeds'))


def _warn_protocol_path(self):
    """The protocol methods build E (x U) Python dicts per step: fine at RLlib's handful of envs per worker, not at E = 65 536."""
    if self.core.E > 1024 and not getattr(self, '_warned_big', False):
        import warnings
        self._warned_big = True
        warnings.warn(f"{type(self).__name__}: num_envs={self.core.E} through the per-env dict protocol costs Python time per env and step; "
                      "use poll_tensors() / send_action_tensor() (device tensors, no per-env objects) for large batches", RuntimeWarning,
                      stacklevel=3)


class _LockStepResets:
    """RLlib resets env copies one by one (`reset_at(i)` / `try_reset(i)`, in whatever order its sampler walks them) when
    they hit the horizon; the batch can only reset as a whole.  The FIRST per-index reset request after a step (or a repeated
    request for an index already served) resets the batch; the other indices are then served from that same reset.  A request
    for the whole batch (`vector_reset()`, index None) ALWAYS resets: with rand_episodes every reset of the reference starts a
    new episode (base.py:169-189), two of them in a row included."""

    def _init_resets(self):
        self._served = set()
        self._stepped = True             # nothing has been reset yet
        self._reset_slot = 2             # persistent views: reset outputs alternate between host buffers 1 and 2 (0 = step outputs)

    def _next_reset_slot(self):
        self._reset_slot = 3 - self._reset_slot
        return self._reset_slot

    def _reset_for(self, index):
        if index is None or self._stepped or index in self._served:
            self.core.reset()
            self._after_core_reset()
            self._served = set()
            self._stepped = False
        if index is not None:
            self._served.add(index)


class CentralVectorEnv(_VectorEnvBase, _LockStepResets):
    """VectorEnv over E central (DeepCoMP) envs.  Actions: list of E int vectors (central.py:28)."""

    def __init__(self, env_config):
        self.core = _core_from_config(env_config, 'central')
        U, B = self.core.U, self.core.B
        self._flat = bool(env_config.get('flat_obs', False))
        if self._flat:       # the flattening of the Dict below, as the space itself: connected U*B | dr U*B | utility U, all inside [-1, 1]
            obs_space = spaces.Box(low=-1, high=1, shape=(U * (2 * B + 1),))
        else:
            obs_space = spaces.Dict({'connected': spaces.MultiBinary(U * B), 'dr': spaces.Box(low=0, high=1, shape=(U * B,)),
                                     'utility': spaces.Box(low=-1, high=1, shape=(U,))})
        _VectorEnvBase.__init__(self, obs_space, spaces.MultiDiscrete([B + 1] * U), self.core.E)
        self._ue_keys = [f'UE {ue}' for ue in env_config['ue_list']]          # the keys of info()'s vector_metrics (base.py:407-408)
        self._all = None
        self._obs = None                 # the observations handed out last (a step's or a reset's)
        self._obs_p = {}                 # persistent views: {host buffer slot: [per-env dict of views]}
        self._persistent = bool(env_config.get('persistent_views', True))
        self._info_level = env_config.get('info_level', 'full')
        if self._info_level not in ('full', 'scalar', 'none'):
            raise ValueError("info_level must be 'full', 'scalar' or 'none'")
        self._dones = [False] * self.core.E
        self._init_resets()

    def _obs_list(self, slot=0):
        """slot 0: a step's outputs; 1 / 2: a reset's (buffers of their own: the last step's views stay valid across a reset)."""
        U, B = self.core.U, self.core.B
        _warn_protocol_path(self)
        if self._persistent:
            self._all = self.core.outputs_host(persistent=True, slot=slot)   # ONE D2H copy into a pinned buffer of the env; the SAME views every time
            if slot not in self._obs_p:                                       # the per-env dicts of views (flat_obs: the rows): built once per buffer, refilled in place
                self._obs_p[slot] = (list(self._all['obs']) if self._flat else
                                     [{'connected': row[:U * B], 'dr': row[U * B:2 * U * B], 'utility': row[2 * U * B:]} for row in self._all['obs']])
            self._obs = self._obs_p[slot]
            return self._obs
        self._all = self.core.outputs_host()                           # ONE D2H copy of obs + reward + info; below are views
        host = self._all['obs']
        self._obs = list(host) if self._flat else [{'connected': row[:U * B], 'dr': row[U * B:2 * U * B], 'utility': row[2 * U * B:]} for row in host]
        return self._obs

    def _after_core_reset(self):
        self._obs_list(self._next_reset_slot())

    def vector_reset(self):
        self._reset_for(None)                    # always a fresh reset; the per-index requests that follow get a new one too
        self._stepped = True
        return self._obs

    def reset_at(self, index=None):
        index = 0 if index is None else int(index)
        self._reset_for(index)
        return self._obs[index]

    def vector_step(self, actions):
        a = torch.from_numpy(np.ascontiguousarray(np.asarray(actions, dtype=np.uint8).reshape(self.core.E, self.core.U)))
        self.core.step(a.to(self.core.device))
        self.core.check()
        self._stepped = True
        obs = self._obs_list()
        rew = self._all['reward'].tolist()
        t, keys = self.core.time, self._ue_keys
        if self._info_level == 'full':                                                                  # base.py:383-411
            su, dr, ut = self._all['sum_utility'].tolist(), self._all['ue_dr'].tolist(), self._all['ue_utility'].tolist()
            infos = [{'time': t, 'scalar_metrics': {'sum_utility': s_}, 'vector_metrics': {'dr': dict(zip(keys, d_)), 'utility': dict(zip(keys, u_))}}
                     for s_, d_, u_ in zip(su, dr, ut)]
        elif self._info_level == 'scalar':
            infos = [{'time': t, 'scalar_metrics': {'sum_utility': s_}} for s_ in self._all['sum_utility'].tolist()]
        else:
            ti = {'time': t}
            infos = [ti] * self.core.E
        return obs, rew, (self._dones if self._persistent else [False] * self.core.E), infos

    # ---- zero-copy path
    def poll_tensors(self):
        """(obs [E, U(2B+1)], reward [E]) device tensors of the last reset / step -- rows are flatten_obs() of the dicts."""
        return self.core.obs, self.core.reward

    def send_action_tensor(self, actions):
        """actions: uint8 [E, U] on the env's device."""
        self.core.step(actions)
        self._stepped = True

    def get_unwrapped(self):
        return [self.core]


class MultiAgentBaseEnv(_BaseEnvBase, _LockStepResets):
    """BaseEnv (poll / send_actions / try_reset) over E multi-agent (DD-/D3-CoMP) envs; agent ids = the UE ids of the
    env_config ('1'..'U', env_setup.py:145-161)."""

    def __init__(self, env_config):
        self.core = _core_from_config(env_config, 'multi')
        B = self.core.B
        self.agent_ids = [str(ue.id) for ue in env_config['ue_list']]
        self.action_space = spaces.Discrete(B + 1)
        self._flat = bool(env_config.get('flat_obs', False))
        if self._flat:       # the flattening of the Dict below, as the space itself: connected | dr | ues_at_bs | util_at_bs | utility
            self.observation_space = spaces.Box(low=-1, high=1, shape=(4 * B + 1,))
        else:
            self.observation_space = spaces.Dict({'connected': spaces.MultiBinary(B), 'dr': spaces.Box(low=0, high=1, shape=(B,)),
                                                  'utility': spaces.Box(low=-1, high=1, shape=(1,)),
                                                  'ues_at_bs': spaces.Box(low=0, high=1, shape=(B,)),
                                                  'util_at_bs': spaces.Box(low=-1, high=1, shape=(B,))})
        self._all = None
        self._reset_obs = None
        self._fresh = True
        self._persistent = bool(env_config.get('persistent_views', True))
        self._view_dict = None
        import operator
        self._pick = operator.itemgetter(*self.agent_ids) if len(self.agent_ids) > 1 else (lambda d, k=self.agent_ids[0]: (d[k],))
        self._dones = {e: {'__all__': False} for e in range(self.core.E)}
        self._no_infos = {e: {} for e in range(self.core.E)}
        self._zeros = {e: dict.fromkeys(self.agent_ids, 0.0) for e in range(self.core.E)}
        self._act_host = np.zeros((self.core.E, self.core.U), dtype=np.uint8)
        self._init_resets()
        self._reset_for(None)

    def _build_views(self, host):
        B = self.core.B
        if self._flat:                            # {env: {agent: row of the pinned array}}
            return {e: dict(zip(self.agent_ids, host[e])) for e in range(self.core.E)}
        return {e: {aid: {'connected': host[e, i, 0:B], 'dr': host[e, i, B:2 * B], 'ues_at_bs': host[e, i, 2 * B:3 * B],
                          'util_at_bs': host[e, i, 3 * B:4 * B], 'utility': host[e, i, 4 * B:4 * B + 1]}
                    for i, aid in enumerate(self.agent_ids)} for e in range(self.core.E)}

    def _views(self, slot=0):
        """slot 0: a step's outputs; 1 / 2: a reset's (buffers of their own: the last step's views stay valid across a reset)."""
        _warn_protocol_path(self)
        if self._persistent:
            self._all = self.core.outputs_host(persistent=True, slot=slot)   # ONE D2H copy into a pinned buffer of the env
            if self._view_dict is None:
                self._view_dict = {}
            if slot not in self._view_dict:                             # {env: {agent: {key: view}}} over that buffer: built once
                self._view_dict[slot] = self._build_views(self._all['obs'])
            return self._view_dict[slot]
        self._all = self.core.outputs_host()                            # ONE D2H copy of obs + reward + info
        return self._build_views(self._all['obs'])                      # [E, U, 4B+1]

    def _after_core_reset(self):
        self._reset_obs = self._views(self._next_reset_slot())

    def poll(self):
        E = self.core.E
        if self._fresh:                                                 # right after a reset: observations only
            self._fresh = False
            if self._persistent:
                return self._reset_obs, self._zeros, self._dones, self._no_infos, {}
            zeros = {e: {a: 0.0 for a in self.agent_ids} for e in range(E)}
            dones = {e: {'__all__': False} for e in range(E)}
            return self._reset_obs, zeros, dones, {e: {} for e in range(E)}, {}
        obs = self._views()
        ids = self.agent_ids
        rewards = dict(enumerate(dict(zip(ids, row)) for row in self._all['reward'].tolist()))
        if self._persistent:
            # `dones` never changes (multi_agent.py:97-99: the horizon ends episodes, not the env); every agent of every env gets the SAME
            # info dict of this step ({'time': t}, multi_agent.py:102-107) -- a new one per step, so nothing aliases across steps
            per_env = dict.fromkeys(ids, {'time': self.core.time})
            return obs, rewards, self._dones, dict.fromkeys(range(E), per_env), {}
        dones = {e: {'__all__': False} for e in range(E)}
        infos = {e: {a: {'time': self.core.time} for a in ids} for e in range(E)}
        return obs, rewards, dones, infos, {}

    def send_actions(self, action_dict):
        a = self._act_host
        a.fill(0)
        for e, acts in action_dict.items():
            try:
                a[e] = self._pick(acts)                                  # every agent acted (RLlib's sampler): one C-level lookup per env
            except KeyError:
                for i, aid in enumerate(self.agent_ids):
                    if aid in acts:                                      # multi_agent.py:30: missing ids are no-ops
                        a[e, i] = int(acts[aid])
        self.core.step(torch.from_numpy(a).to(self.core.device))
        self.core.check()
        self._stepped = True

    def try_reset(self, env_id=None):
        env_id = 0 if env_id is None else int(env_id)
        self._reset_for(env_id)
        return self._reset_obs[env_id]

    # ---- zero-copy path
    def poll_tensors(self):
        """(obs [E, U, 4B+1], reward [E, U]) device tensors of the last reset / step; obs[e, i] is flatten_obs() of agent
        agent_ids[i]'s dict in env e."""
        return self.core.obs, self.core.reward

    def send_action_tensor(self, actions):
        """actions: uint8 [E, U] on the env's device, column i = agent agent_ids[i]."""
        self.core.step(actions)
        self._stepped = True

    def get_unwrapped(self):
        return [self.core]
