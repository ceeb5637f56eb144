"""RLlib-facing batched adapters.

RLlib 1.4 (the reference's pinned version, setup.py:14) steps several env copies per rollout worker through its
``VectorEnv`` protocol -- ``vector_reset() / reset_at(i) / vector_step(actions) / get_unwrapped()`` -- and
multi-agent envs through ``BaseEnv.poll() / send_actions()``.  ``CentralVectorEnv`` and ``MultiAgentBaseEnv`` expose ONE
``BatchedMobileEnv`` (E envs, one kernel launch per step) through those protocols: one device->host copy of the packed
observation tensor per step, then per-env *views* (no per-env device work).  The observation dicts have the keys and
shapes of the reference's spaces (central.py:147-151, variants.py:255-269), so the untouched PPO config's preprocessor
flattens them in the same sorted-key order the packed tensor already uses.

``ray`` is not installed in the build image: the classes derive from RLlib's base classes when importable and are plain
duck-typed classes otherwise; tests/test_parity_gpu.py checks them against E independent single-env instances.
All envs of a batch run in lock step (shared ``time``), which is how RLlib drives a VectorEnv with a fixed horizon.
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


def _core_from_config(env_config, kind):
    return BatchedMobileEnv(env_config['map'], env_config['bs_list'], env_config['ue_list'], kind,
                            num_envs=int(env_config.get('num_envs', 1)), seed=env_config['seed'],
                            episode_length=env_config['episode_length'], reward=env_config['reward'],
                            rand_episodes=env_config['rand_episodes'], rng=env_config.get('rng', 'philox'),
                            device=env_config.get('device', 'cuda'), env_id_base=env_config.get('env_id_base', 0))


class CentralVectorEnv(_VectorEnvBase):
    """VectorEnv over E central (DeepCoMP) envs.  Actions: list of E int vectors (central.py:28)."""

    def __init__(self, env_config):
        self.core = _core_from_config(env_config, 'central')
        U, B = self.core.U, self.core.B
        obs_space = spaces.Dict({'connected': spaces.MultiBinary(U * B), 'dr': spaces.Box(low=0, high=1, shape=(U * B,)),
                                 'utility': spaces.Box(low=-1, high=1, shape=(U,))})
        super().__init__(obs_space, spaces.MultiDiscrete([B + 1] * U), self.core.E)
        self._host = None

    def _obs_list(self):
        U, B = self.core.U, self.core.B
        self._all = self.core.outputs_host()                           # ONE D2H copy of obs + reward + info; below are views
        self._host = self._all['obs']
        return [{'connected': row[:U * B], 'dr': row[U * B:2 * U * B], 'utility': row[2 * U * B:]} for row in self._host]

    def vector_reset(self):
        self.core.reset()
        return self._obs_list()

    def reset_at(self, index=None):
        """Lock-step batch: resetting one env resets the episode of all (RLlib calls this at the shared horizon)."""
        if index in (None, 0):
            self.core.reset()
            self._obs_cache = self._obs_list()
        return self._obs_cache[index or 0]

    def vector_step(self, actions):
        a = torch.from_numpy(np.ascontiguousarray(np.asarray(actions, dtype=np.uint8).reshape(self.core.E, self.core.U)))
        self.core.step(a.to(self.core.device))
        self.core.check()
        obs = self._obs_list()
        rew, su = self._all['reward'].tolist(), self._all['sum_utility'].tolist()
        t = self.core.time
        infos = [{'time': t, 'scalar_metrics': {'sum_utility': su[e]}} for e in range(self.core.E)]
        return obs, rew, [False] * self.core.E, infos

    def get_unwrapped(self):
        return [self.core]


class MultiAgentBaseEnv(_BaseEnvBase):
    """BaseEnv (poll / send_actions) over E multi-agent (DD-/D3-CoMP) envs; agent ids '1'..'U' (env_setup.py:145-161)."""

    def __init__(self, env_config):
        self.core = _core_from_config(env_config, 'multi')
        B = self.core.B
        self.agent_ids = [str(ue.id) for ue in env_config['ue_list']]
        self.action_space = spaces.Discrete(B + 1)
        self.observation_space = spaces.Dict({'connected': spaces.MultiBinary(B), 'dr': spaces.Box(low=0, high=1, shape=(B,)),
                                              'utility': spaces.Box(low=-1, high=1, shape=(1,)),
                                              'ues_at_bs': spaces.Box(low=0, high=1, shape=(B,)),
                                              'util_at_bs': spaces.Box(low=-1, high=1, shape=(B,))})
        self._pending = None
        self._fresh = True
        self.core.reset()

    def _views(self):
        B = self.core.B
        self._all = self.core.outputs_host()                            # ONE D2H copy of obs + reward + info
        host = self._all['obs']                                         # [E, U, 4B+1]
        return {e: {aid: {'connected': host[e, i, 0:B], 'dr': host[e, i, B:2 * B], 'ues_at_bs': host[e, i, 2 * B:3 * B],
                          'util_at_bs': host[e, i, 3 * B:4 * B], 'utility': host[e, i, 4 * B:4 * B + 1]}
                    for i, aid in enumerate(self.agent_ids)} for e in range(self.core.E)}

    def poll(self):
        obs = self._views()
        E = self.core.E
        if self._fresh:
            self._fresh = False
            zeros = {e: {a: 0.0 for a in self.agent_ids} for e in range(E)}
            dones = {e: {'__all__': False} for e in range(E)}
            return obs, zeros, dones, {e: {} for e in range(E)}, {}
        rew = self._all['reward'].tolist()
        rewards = {e: dict(zip(self.agent_ids, rew[e])) for e in range(E)}
        dones = {e: {'__all__': False} for e in range(E)}
        infos = {e: {a: {'time': self.core.time} for a in self.agent_ids} for e in range(E)}
        return obs, rewards, dones, infos, {}

    def send_actions(self, action_dict):
        a = np.zeros((self.core.E, self.core.U), dtype=np.uint8)
        for e, acts in action_dict.items():
            for i, aid in enumerate(self.agent_ids):
                if aid in acts:                                          # multi_agent.py:30: missing ids are no-ops
                    a[e, i] = int(acts[aid])
        self.core.step(torch.from_numpy(a).to(self.core.device))
        self.core.check()

    def try_reset(self, env_id=None):
        if env_id in (None, 0):
            self.core.reset()
            self._reset_views = self._views()
        return self._reset_views[env_id or 0]

    def get_unwrapped(self):
        return [self.core]
