"""MI355X-native DeepCoMP environments behind the reference's gym / RLlib surface.

``BatchedMobileEnv`` is the engine: E independent envs stepped by one fused HIP kernel launch, state and
observations living in device tensors (``pos[E*U,2] f64``, ``mv[E*U] i64``, ``conn[E*U] i32``, ``ewma[E*U] f32``).

``CentralRelNormEnv`` and ``MultiAgentMobileEnv`` keep the reference's class names, constructor
(``env_config`` dict, env_setup.py:247-256), ``reset()``, ``step(action)``, ``seed()`` and attribute surface
(deepcomp/env/multi_ue/central.py:143-152, multi_ue/multi_agent.py:6-107, single_ue/base.py:20-466):
* ``num_envs == 1`` (default): return values have the reference's exact Python shapes (dict observations, float /
  dict rewards, ``done=None``, ``info`` dict), so RLlib's rollout worker or ``Simulation.run_episode`` can drive
  it unchanged;
* ``num_envs > 1`` (extra ``env_config['num_envs']`` key): ``step`` takes an ``[E, U]`` action tensor and returns
  device tensors (zero-copy for a learner on the same GPU).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib, rng as _rng, spaces

# The reference's classes ARE gym / RLlib classes (single_ue/base.py:20 `class MobileEnv(gym.Env)`, multi_ue/multi_agent.py:6
# `class MultiAgentMobileEnv(RelNormEnv, MultiAgentEnv)`), and its callers dispatch on that: `MultiAgentEnv in
# env_class.__mro__` (util/env_setup.py:289, util/simulation.py:46) decides whether RLlib gets a policy map, and RLlib's own
# BaseEnv conversion tests isinstance(env, MultiAgentEnv).  So the drop-in classes derive from the REAL base classes whenever
# gym / ray are importable; without them (this build image has neither) neutral stand-ins keep the module importable.
try:                                            # pragma: no cover - gym is absent in the build image
    from gym import Env as _GymEnv
    HAVE_GYM = True
except Exception:                               # noqa: BLE001
    HAVE_GYM = False

    class _GymEnv:                              # attribute surface of gym.Env that callers read
        metadata = {'render.modes': []}
        reward_range = (-float('inf'), float('inf'))
        spec = None
        action_space = None
        observation_space = None

        def close(self):
            pass

        @property
        def unwrapped(self):
            return self

try:                                            # pragma: no cover - ray is absent in the build image
    from ray.rllib.env.multi_agent_env import MultiAgentEnv as _MultiAgentEnv
    HAVE_RAY = True
except Exception:                               # noqa: BLE001
    HAVE_RAY = False

    class _MultiAgentEnv:                       # RLlib's MultiAgentEnv has no __init__ either (multi_agent.py:14)
        pass

SNR_THRESHOLD = 2e-8          # station.py:10
MIN_UTILITY, MAX_UTILITY = -20, 20   # constants.py:40-41


def _fragment_obs(out):
    """The observation buffer of a rollout's `out` dict: the rows, or the compact record where the caller asked for that."""
    return out['obs'] if out.get('obs') is not None else out.get('obs_compact')


def _coord(v):
    return -1 if v == 'random' else int(v)


def parse_entities(map, bs_list, ue_list):
    """Objects of an env_config (env_setup.py:247-256) -> flat config arrays.  Duck-typed on purpose: works with
    deepcomp_amd.entities AND with the reference's own Map / Basestation / User / RandomWaypoint objects
    (map.py:20-21, station.py:16-19, user.py:27-40, movement.py:96-97), which is what makes the classes below drop-in
    behind the reference's unchanged scenario factory."""
    for bs in bs_list:
        assert bs.sharing_model in _lib.SHARING, f"{bs.sharing_model=} not supported."       # station.py:22
    for ue in ue_list:
        if ue.util_func not in _lib.UTILITY:
            raise NotImplementedError(f"Utility function {ue.util_func} not implemented!")   # user.py:92
    pause, border = [], []
    for ue in ue_list:                       # RandomWaypoint(map, velocity, pause_duration=2, border_buffer=10), movement.py:87-104
        mvt = ue.movement
        pd, bb = getattr(mvt, 'pause_duration', 2), getattr(mvt, 'border_buffer', 10)
        v = mvt.init_velocity
        if not isinstance(v, str) and not 0 <= float(v) <= 1e6:
            raise NotImplementedError(f"UE {ue.id}: velocity {v} outside 0..1e6")
        if int(pd) != pd or int(bb) != bb or not 0 <= pd <= 127 or not 1 <= bb <= 255:
            raise NotImplementedError(f"UE {ue.id}: pause_duration={pd} / border_buffer={bb}: integers in 0..127 / 1..255 are implemented")
        pause.append(int(pd)); border.append(int(bb))
    vel_specs = [ue.movement.init_velocity for ue in ue_list]
    # movement.py:116-117: a fixed velocity is whatever number the caller passed.  Integers in 0..255 live in the movement word;
    # anything else (2.5, 300) goes to the device as a per-UE number (dcomp_cfg.ue_velocity) and the word's field stays 0.
    vel_num = np.array([-1.0 if isinstance(v, str) or (float(v) == int(v) and int(v) <= 255) else float(v) for v in vel_specs], dtype=np.float64)
    vr = [_rng.vel_range(v) if n < 0 else (0, 0) for v, n in zip(vel_specs, vel_num)]
    init_xy = [(_coord(ue.init_pos_x), _coord(ue.init_pos_y)) for ue in ue_list]
    return {
        'map_w': int(map.width), 'map_h': int(map.height),
        'bs_x': np.array([float(bs.pos.x) for bs in bs_list], dtype=np.float64),
        'bs_y': np.array([float(bs.pos.y) for bs in bs_list], dtype=np.float64),
        'bs_sharing': np.array([_lib.SHARING[bs.sharing_model] for bs in bs_list], dtype=np.int32),
        'ue_ids': [ue.id for ue in ue_list],
        'ue_util': np.array([_lib.UTILITY[ue.util_func] for ue in ue_list], dtype=np.int32),
        'ue_dr_req': np.array([float(ue.dr_req) for ue in ue_list], dtype=np.float32),
        'vel_specs': [v if n < 0 else 0 for v, n in zip(vel_specs, vel_num)],      # what the draw tapes see: a fixed 0, never drawn
        'vel_num': vel_num,
        'vel_lo': np.array([r[0] for r in vr], dtype=np.int32), 'vel_hi': np.array([r[1] for r in vr], dtype=np.int32),
        'init_xy': init_xy,
        'init_x': np.array([p[0] for p in init_xy], dtype=np.int32), 'init_y': np.array([p[1] for p in init_xy], dtype=np.int32),
        'pause': np.array(pause, dtype=np.int32), 'border': np.array(border, dtype=np.int32),
    }


class BatchedMobileEnv:
    """E lock-stepped envs on one GPU.  All heavy lifting is in libdcomp_hip.so (include/dcomp.h)."""

    def __init__(self, map, bs_list, ue_list, kind, num_envs=1, seed=42, episode_length=100, reward='avg',
                 rand_episodes=False, rng='philox', device='cuda', env_id_base=0, env_seeds=None, log_metrics=True,
                 tape_depth=None, ue_arrival=None, new_ue_interval=None, max_ues=None, host_io=False):
        L = _lib.load()
        # host_io: outputs and actions live in PINNED HOST memory that the kernels read / write directly over PCIe (a few
        # hundred bytes per step at num_envs = 1): the single-env compatibility classes then need no copy calls at all,
        # only the stream synchronisation of check().  Batched envs keep everything in HBM.
        self.host_io = bool(host_io)
        self._L = L
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError("deepcomp_amd runs on an AMD GPU (torch device 'cuda'); there is no CPU path")
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.kind = _lib.MULTI if kind in ('multi', _lib.MULTI) else _lib.CENTRAL
        # UE arrival / departure (base.py:52-56, 433-443): U = slots per env = max_ues (base.py:79-84), U0 = initial list
        self.ue_arrival = {int(k): int(v) for k, v in ue_arrival.items()} if ue_arrival else None
        self.new_ue_interval = None if self.ue_arrival else new_ue_interval
        self.U0 = len(ue_list)
        self.dynamic = bool(self.ue_arrival) or self.new_ue_interval is not None
        auto_max = _rng.max_num_ue(self.U0, int(episode_length), self.ue_arrival, self.new_ue_interval)
        cap = int(max_ues) if max_ues else auto_max
        assert cap >= self.U0                                                         # base.py:84
        if self.dynamic and cap < auto_max:
            raise ValueError(f"max_ues={cap} is smaller than the schedule needs ({auto_max})")
        self.schedule = _rng.arrival_schedule(int(episode_length), self.ue_arrival, self.new_ue_interval) if self.dynamic else None
        self.max_id = self.U0 + (sum(a for _, a in self.schedule) if self.dynamic else 0)
        self.E, self.U, self.B = int(num_envs), (cap if self.dynamic else self.U0), len(bs_list)
        self.map_w, self.map_h = int(map.width), int(map.height)
        self.episode_length = int(episode_length)
        self.rand_episodes = bool(rand_episodes)
        self.log_metrics = bool(log_metrics)
        self._policy_key, self.next_action = None, None      # in-step heuristic policy (set_policy)
        self.want_reward_before = False      # per-UE pre-move reward (single-agent env), off by default: 4 B/UE of extra traffic
        if reward not in _lib.REWARD:
            raise NotImplementedError(f"Unexpected reward aggregation: {reward}")       # central.py:73
        self.reward_agg = reward
        self.rng_mode = _lib.RNG_PHILOX if rng == 'philox' else _lib.RNG_TAPE
        if rng not in ('philox', 'reference'):
            raise ValueError("rng must be 'philox' (counter-based, in-kernel) or 'reference' (stdlib-random draw tape)")
        self._tape_depth_arg = tape_depth
        self.seed_value = seed if seed is not None else int(np.random.SeedSequence().generate_state(1)[0])
        self.env_id_base = int(env_id_base)
        # env e of a batch gets base seed `seed + 20000*e` (UE i adds 100*(i+1), base.py:138-143)
        # (UE offsets reach 100*U: with more than 199 UEs a stride of 20000 would make env e's UE i+200 replay env e+1's UE i)
        self._seed_stride = max(20000, 100 * (len(ue_list) + 1))
        self._reseeded = False
        self._device_seed = self.seed_value    # the Philox key the DEVICE draws from (a seed() call stages a new one until reset())
        self.env_seeds = (np.asarray(env_seeds, dtype=np.int64) if env_seeds is not None
                          else self.seed_value + self._seed_stride * (self.env_id_base + np.arange(self.E, dtype=np.int64)))

        U, B = self.U, self.B          # U: slots per env
        ent = parse_entities(map, bs_list, ue_list)
        self._bs_x, self._bs_y, self._bs_sh = ent['bs_x'], ent['bs_y'], ent['bs_sharing']
        self._ue_util, self._ue_req = ent['ue_util'], ent['ue_dr_req']
        self.vel_specs, self._vlo, self._vhi = ent['vel_specs'], ent['vel_lo'], ent['vel_hi']
        self.init_xy, self._ix, self._iy = ent['init_xy'], ent['init_x'], ent['init_y']
        self._pause, self._border = ent['pause'], ent['border']
        self._vel_num = ent['vel_num'] if (ent['vel_num'] >= 0).any() else None
        # A UE redraws (velocity, waypoint) at most once per pause_duration + 1 steps: it arrives, stands still for pause_duration
        # steps and draws in the step it moves on (movement.py:158-181) -- every third step with the default pause of 2, EVERY
        # step with pause_duration 0 and a velocity that covers the distance.  UEs that arrive during an episode pause 2 steps.
        self._redraw_period = min([int(x) for x in self._pause] + ([2] if self.dynamic else [])) + 1
        self.tape_depth = int(self._tape_depth_arg or (self.episode_length // self._redraw_period + 4))

        c = _lib.DcompCfg()
        c.num_envs, c.num_ue, c.num_bs = self.E, self.U0, B
        c.max_ues = self.U if self.dynamic else 0
        c.map_w, c.map_h = self.map_w, self.map_h
        c.env_kind, c.reward_agg = self.kind, _lib.REWARD[reward]
        c.rng_mode, c.tape_depth = self.rng_mode, self.tape_depth
        c.device = self.device.index
        c.seed = int(self.seed_value) & 0xFFFFFFFFFFFFFFFF
        c.env_id_base = self.env_id_base
        dp, ip, fp = _lib._dp, _lib._ip, _lib._fp
        c.bs_x, c.bs_y = self._bs_x.ctypes.data_as(dp), self._bs_y.ctypes.data_as(dp)
        c.bs_sharing = self._bs_sh.ctypes.data_as(ip)
        c.ue_util, c.ue_dr_req = self._ue_util.ctypes.data_as(ip), self._ue_req.ctypes.data_as(fp)
        c.ue_vel_lo, c.ue_vel_hi = self._vlo.ctypes.data_as(ip), self._vhi.ctypes.data_as(ip)
        c.ue_init_x, c.ue_init_y = self._ix.ctypes.data_as(ip), self._iy.ctypes.data_as(ip)
        c.ue_pause_duration, c.ue_border_buffer = self._pause.ctypes.data_as(ip), self._border.ctypes.data_as(ip)
        if self._vel_num is not None:
            c.ue_velocity = self._vel_num.ctypes.data_as(dp)
        self._cfg = c
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.create(c, self._h))

        n, dev = self.E * U, self.device
        self.pos = torch.zeros((n, 2), dtype=torch.float64, device=dev)
        self.mv = torch.zeros(n, dtype=torch.int64, device=dev)
        self.conn = torch.zeros(n, dtype=torch.int32, device=dev)
        # more than 32 stations (or more than 256 UE slots per env): the generic kernel; stations 32-63 of the connection sets in a second word per UE (dcomp_state.conn_hi)
        # (every env of the generic kernel has the word; the LIBRARY says whether this env is one -- it knows every reason to take that kernel)
        needs_hi = (L.dcomp_needs_conn_hi(self._h) == 1) if hasattr(L, 'dcomp_needs_conn_hi') else (B > _lib.MASK32_MAX_BS or U > _lib.SPECIAL_MAX_UE)
        self.conn_hi = torch.zeros(n, dtype=torch.int32, device=dev) if needs_hi else None
        self.ewma = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flags = torch.zeros(4, dtype=torch.int32, device=dev)
        # all outputs of one step live in ONE flat buffer (sections 16-byte aligned): the single-env compatibility mode
        # fetches everything a step returns with a single device->host copy
        self.obs_dim = 4 * B + 1 if self.kind == _lib.MULTI else U * (2 * B + 1)
        obs_shape = (self.E, U, self.obs_dim) if self.kind == _lib.MULTI else (self.E, self.obs_dim)
        rew_shape = (self.E, U) if self.kind == _lib.MULTI else (self.E,)
        sections = [('obs', obs_shape), ('reward', rew_shape), ('sum_utility', (self.E,)), ('ue_dr', (self.E, U)),
                    ('ue_utility', (self.E, U)), ('reward_before', (self.E, U))]
        off, self._sections = 0, {}
        for name, shape in sections:
            cnt = int(np.prod(shape))
            self._sections[name] = (off, cnt, shape)
            off += (cnt + 3) // 4 * 4
        self._outbuf = torch.zeros(off, dtype=torch.float32).pin_memory() if self.host_io else torch.zeros(off, dtype=torch.float32, device=dev)
        self.action_host = torch.zeros((self.E, U), dtype=torch.uint8).pin_memory() if self.host_io else None
        for name, (o, cnt, shape) in self._sections.items():
            setattr(self, name, self._outbuf[o:o + cnt].view(shape))
        since_bytes = ctypes.c_size_t(0)
        _lib.check(L.dcomp_state_sizes(self._h, None, None, None, None, None, ctypes.byref(since_bytes)))
        self.conn_since = torch.zeros(since_bytes.value // 2, dtype=torch.int16, device=dev) if since_bytes.value else None
        self.uid = torch.zeros(n, dtype=torch.int16, device=dev) if self.dynamic else None
        self.orig_consumed = torch.zeros(self.E * self.U0, dtype=torch.int16, device=dev) if self.dynamic else None
        self._st = _lib.DcompState(self.pos.data_ptr(), self.mv.data_ptr(), self.conn.data_ptr(), self.ewma.data_ptr(),
                                   self.flags.data_ptr(), self.conn_since.data_ptr() if self.conn_since is not None else None,
                                   self.uid.data_ptr() if self.dynamic else None,
                                   self.orig_consumed.data_ptr() if self.dynamic else None,
                                   self.conn_hi.data_ptr() if self.conn_hi is not None else None)
        self._dyn_streams = None
        self._ev_keep = None
        self._out = self._make_out(self.obs, self.reward)
        self._st_ref, self._out_ref = ctypes.byref(self._st), ctypes.byref(self._out)   # per-step ctypes work done once
        self._tape_dev = None
        self._streams = None
        self._fixed_tape = None
        self._live = None                    # seed(immediate=True) in a running fixed episode: (new env seeds, cursors at that time)

    # ------------------------------------------------------------------ helpers
    def _make_out(self, obs, reward, packed=None):
        m = self.log_metrics
        return _lib.DcompOut(obs.data_ptr() if obs is not None else None, reward.data_ptr(), self.sum_utility.data_ptr() if m else None,
                             self.ue_dr.data_ptr() if m else None, self.ue_utility.data_ptr() if m else None,
                             self.reward_before.data_ptr() if self.want_reward_before else None,
                             packed.data_ptr() if packed is not None else None)

    def _stream(self):
        try:                                   # raw handle of torch's current stream without building a Stream object (~4 us)
            return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(self.device.index))
        except AttributeError:                 # pragma: no cover - private torch API moved
            return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def __del__(self):
        h = getattr(self, '_h', None)
        if h is not None and h.value:
            self._L.dcomp_destroy(h)
            self._h = None

    @property
    def time(self):
        return self._L.dcomp_time(self._h)

    @property
    def episode(self):
        return self._L.dcomp_episode(self._h)

    def seed(self, seed=None, immediate=False):
        """MobileEnv.seed (base.py:132-143); None leaves the generators alone.

        immediate=False (the batched API): the gym convention `env.seed(s); env.reset()` -- the new seed is STAGED here and
        governs every draw from the next reset() on; the remaining draws of an episode in progress keep coming from the tape /
        Philox key the episode started with, in both RNG modes.

        immediate=True (what the reference-named classes pass in rng='reference' mode): the reference to the letter.  Both
        streams of every UE are re-seeded AT ONCE, so the waypoints / velocities a UE still draws in the running episode come
        from the start of its new movement stream (base.py:138-143 -> user.py:94-96); the next reset() of a
        rand_episodes=False env re-seeds with the CONFIGURED seed again (base.py:171-173: seed() then only shapes the rest of
        the running episode), a rand_episodes=True env carries the new streams on.  Held against reference-run trajectories
        (tests/golden/reseed_*.npz, reseeddyn_*.npz).  With UE arrival / departure the UEs of the CURRENT list -- arrived ones
        too -- are re-seeded by list position, the departure / arrival-point generators restart, later arrivals keep the
        configured seed (base.py:601-604)."""
        if seed is None:
            return
        seed = int(seed)
        if immediate and self.rng_mode == _lib.RNG_TAPE:
            return self._seed_live(seed)
        self.seed_value = seed
        self.env_seeds = self.seed_value + self._seed_stride * (self.env_id_base + np.arange(self.E, dtype=np.int64))
        self._streams, self._fixed_tape, self._dyn_streams = None, None, None
        self._live = None
        self._reseeded = True                         # reset() starts the new key's episode 0

    def _seed_live(self, seed):
        """seed(immediate=True), rng='reference': splice the new streams into the tape of the running episode."""
        seeds = seed + self._seed_stride * (self.env_id_base + np.arange(self.E, dtype=np.int64))
        live = self._tape_dev is not None
        if self.dynamic:
            if self._dyn_streams is None or not live:        # no episode yet
                if self.rand_episodes:                       # ... the first reset() starts the new streams; a fixed-episode env re-seeds
                    self.seed_value, self.env_seeds = seed, seeds        #     with the configured seed anyway (base.py:171-173)
                return
            E, U, n = self.E, self.U, self.num_ue
            uid = self.uid.cpu().numpy().astype(np.uint16).reshape(E, U)
            cur = ((self.mv >> 48) & 0xFFFF).cpu().numpy().reshape(E, U)
            lists = [[(int(w & 0x7FFF), bool(w & 0x8000)) for w in uid[e, :n]] for e in range(E)]
            pos0, trip = self._dyn_streams.reseed_live(seeds, lists, cur)
            torch.cuda.current_stream(self.device).synchronize()       # steps in flight still read the old tape
            old = self._tape_dev
            tape = self._upload_tape(pos0, trip)
            _lib.check(self._L.dcomp_set_tape(self._h, ctypes.byref(tape), int(trip.shape[1])))
            del old
            return
        cursor = ((self.mv >> 48) & 0xFFFF).cpu().numpy().astype(np.int64) if live else None   # triples consumed so far, per (env, UE)
        if self.rand_episodes:
            self.seed_value, self.env_seeds = seed, seeds
            if self._streams is None:                 # no reset() yet: the first one starts the new streams
                return
            spliced = self._streams.reseed_live(seeds, cursor)
            if spliced is None:
                return
        else:                                         # reset() goes back to self.env_seeds (base.py:171-173): only the running episode changes
            if not live:
                return
            self._live = (seeds, cursor)
            spliced = self._live_fixed_tape(max(self._tape_depth_now, self.tape_depth))
        pos0, trip = spliced
        torch.cuda.current_stream(self.device).synchronize()       # steps in flight still read the old tape
        old = self._tape_dev
        tape = self._upload_tape(pos0, trip)
        _lib.check(self._L.dcomp_set_tape(self._h, ctypes.byref(tape), int(trip.shape[1])))
        del old

    def _live_fixed_tape(self, depth):
        """rand_episodes=False after a live re-seed: row (e, i) = the configured seed's triples up to the cursor at the time of
        seed(), the new seed's stream from there on (both stateless: the C++ MT19937 draws any depth again)."""
        pos0, trip = _rng.mt_tape(self._cfg, self.env_seeds, depth)
        seeds, cursor = self._live
        _, fresh = _rng.mt_tape(self._cfg, seeds, depth)
        for i, c in enumerate(cursor):
            c = int(c)
            if c < depth:
                trip[i, c:] = fresh[i, :depth - c]
        return pos0, trip

    def _draw_tape_dynamic(self):
        """Reference-exact draws with UE arrival: see rng.DynamicStdlibStreams (initial UEs keep their generators)."""
        first = self._dyn_streams is None
        if first:
            self._dyn_streams = _rng.DynamicStdlibStreams(self.env_seeds, self.map_w, self.map_h, self.vel_specs, self.init_xy,
                                                         self.tape_depth, self.rand_episodes, self.max_id, border=self._border)
            return self._dyn_streams.draw_episode()
        E, U, U0 = self.E, self.U, self.U0
        uid = self.uid.cpu().numpy().astype(np.uint16).reshape(E, U)
        cursor = ((self.mv >> 48) & 0xFFFF).cpu().numpy().reshape(E, U)
        left = self.orig_consumed.cpu().numpy().astype(np.uint16).reshape(E, U0)
        n = self.num_ue
        end_lists, consumed = [], []
        for e in range(E):
            end_lists.append([(int(w & 0x7FFF), bool(w & 0x8000)) for w in uid[e, :n]])
            c = [int(x) for x in left[e]]                       # 0xFFFF: never left -> still listed, cursor of its slot
            for slot in range(n):
                w = int(uid[e, slot])
                if not (w & 0x8000):
                    c[(w & 0x7FFF) - 1] = int(cursor[e, slot])
            consumed.append(c)
        return self._dyn_streams.draw_episode(end_lists, consumed)

    def _draw_tape(self):
        if self.dynamic:
            return self._draw_tape_dynamic()
        if self.rand_episodes:
            if self._streams is None:
                self._streams = _rng.StdlibStreams(self.env_seeds, self.map_w, self.map_h, self.vel_specs, self.init_xy,
                                                   self.tape_depth, border=self._border)
                consumed = None
            else:
                consumed = ((self.mv >> 48) & 0xFFFF).cpu().numpy()
            return self._streams.draw_episode(reseed=False, consumed=consumed)
        self._live = None
        if self._fixed_tape is None or self._fixed_tape[1].shape[1] < self.tape_depth:   # (a live re-seed extended only the spliced tape)
            # re-seeded at every reset (base.py:171-173): same tape every episode
            self._fixed_tape = _rng.mt_tape(self._cfg, self.env_seeds, self.tape_depth)
        return self._fixed_tape

    def _upload_tape(self, pos0, trip):
        self._tape_dev = (torch.from_numpy(pos0).to(self.device), torch.from_numpy(trip.view(np.int16)).to(self.device))
        self._tape_depth_now = int(trip.shape[1])
        return _lib.DcompTape(self._tape_dev[0].data_ptr(), self._tape_dev[1].data_ptr(), self.U0 + self.max_id if self.dynamic else 0)

    def _ensure_tape(self, steps=1):
        """rng='reference': the episode's draw tape must cover the next `steps` steps.  A UE redraws at most once per
        (smallest configured pause_duration + 1) steps (arrive, pause, draw when it moves on: movement.py:158-181), so step t
        needs at most t // period + 2 triples.  Episodes that outlive the tape -- the reference's done() is always None and
        --cont-train never resets (main.py:48-51) -- get a longer one: the same draws, continued."""
        if self.rng_mode != _lib.RNG_TAPE or self._tape_dev is None:
            return
        need = (self.time + steps) // self._redraw_period + 3
        if need <= self._tape_depth_now:
            return
        depth = max(need, 2 * self._tape_depth_now)
        if self.dynamic:
            pos0, trip = self._dyn_streams.extend(depth)
        elif self.rand_episodes:
            pos0, trip = self._streams.extend(depth)
        elif self._live is not None:
            pos0, trip = self._live_fixed_tape(depth)
        else:
            pos0, trip = self._fixed_tape = _rng.mt_tape(self._cfg, self.env_seeds, depth)     # stateless: same prefix, longer
        torch.cuda.current_stream(self.device).synchronize()       # steps in flight still read the old tape
        old = self._tape_dev
        tape = self._upload_tape(pos0, trip)
        _lib.check(self._L.dcomp_set_tape(self._h, ctypes.byref(tape), depth))
        self.tape_depth = depth
        del old

    # ------------------------------------------------------------------ gym-like batched API
    def reset(self, _out=None):
        """MobileEnv.reset (base.py:169-189) for all envs -> first observation tensor."""
        with torch.cuda.device(self.device):
            tape = None
            if self.rng_mode == _lib.RNG_TAPE:
                pos0, trip = self._draw_tape()
                tape = self._upload_tape(pos0, trip)
            else:
                if self._reseeded:                        # seed() since the last reset: the new Philox key starts here
                    _lib.check(self._L.dcomp_set_seed(self._h, ctypes.c_uint64(int(self.seed_value) & 0xFFFFFFFFFFFFFFFF)))
                if not self.rand_episodes or self._reseeded:
                    self._L.dcomp_set_episode(self._h, 0)     # fixed episodes: same Philox counter word every reset
            self._reseeded = False
            self._device_seed = self.seed_value            # the seed this episode's draws come from (tape drawn / key installed above)
            _lib.check(self._L.dcomp_reset(self._h, ctypes.byref(self._st), ctypes.byref(tape) if tape else None,
                                           ctypes.byref(self._out if _out is None else _out), self._stream()))
        if self._policy_key is not None:
            self._policy_launched()
        return self.obs if _out is None else None

    def step(self, action):
        """MobileEnv.step (base.py:413-466).  action: uint8 tensor [E, U] on this device, values in [0, B]."""
        on_dev = action.device == self.device or (self.host_io and action.data_ptr() == self.action_host.data_ptr())
        if action.dtype != torch.uint8 or not on_dev or not action.is_contiguous() or action.numel() != self.E * self.U:
            raise ValueError(f"action must be a contiguous uint8 tensor with {self.E}x{self.U} entries on {self.device}")
        if torch.cuda.current_device() == self.device.index:      # the common case: no device switch around the launch
            self._launch_step(action, self._out)
        else:
            with torch.cuda.device(self.device):
                self._launch_step(action, self._out)
        return self.obs, self.reward, None, self.info()

    def _launch_step(self, action, out):
        if self.rng_mode == _lib.RNG_TAPE:
            self._ensure_tape(1)
        if not self.dynamic:
            rc = self._L.dcomp_step(self._h, self._st_ref, ctypes.c_void_p(action.data_ptr()),
                                    self._out_ref if out is self._out else ctypes.byref(out), self._stream())
            if rc:
                _lib.check(rc)
            if self._policy_key is not None:
                self._policy_launched()
            return
        t = self.time
        n_rem, n_add = self.schedule[t] if t < len(self.schedule) else (0, 0)      # base.py:433-443
        ev = _lib.DcompEvents(n_rem, n_add, None, None)
        if (n_rem or n_add) and self.rng_mode == _lib.RNG_TAPE:
            rem = torch.from_numpy(self._dyn_streams.departures(n_rem, self.num_ue)).to(self.device) if n_rem else None
            add = torch.from_numpy(self._dyn_streams.arrivals(n_add)).to(self.device) if n_add else None
            self._ev_keep = (rem, add)                         # keep the device buffers alive until the launch has run
            ev = _lib.DcompEvents(n_rem, n_add, rem.data_ptr() if n_rem else None, add.data_ptr() if n_add else None)
        _lib.check(self._L.dcomp_step_dyn(self._h, ctypes.byref(self._st), ctypes.c_void_p(action.data_ptr()),
                                          ctypes.byref(out), ctypes.byref(ev), self._stream()))
        if self._policy_key is not None:
            self._policy_launched()

    @property
    def num_ue(self):
        """UEs currently in every env's list (the arrival schedule is configuration, identical for all envs)."""
        return self._L.dcomp_num_ue(self._h)

    def _require(self, t, dtype, numel, what):
        """The kernels get raw pointers: a wrong dtype / device / size would read or write out of bounds on the device."""
        if not isinstance(t, torch.Tensor) or t.dtype != dtype or t.device != self.device or not t.is_contiguous() or t.numel() != numel:
            raise ValueError(f"{what} must be a contiguous {dtype} tensor with {numel} elements on {self.device}")

    def step_into(self, action, obs, reward):
        """Like step() but writes observation / reward into caller-provided tensors (rollout buffers)."""
        self._require(action, torch.uint8, self.E * self.U, 'action')
        self._require(obs, torch.float32, self.obs.numel(), 'obs')
        self._require(reward, torch.float32, self.reward.numel(), 'reward')
        out = self._make_out(obs, reward)
        with torch.cuda.device(self.device):
            self._launch_step(action, out)

    @property
    def compact_words(self):
        """int32 words of one env-step in the compact record (deepcomp_amd.fragment: U (B + 2) + 2B); multi-agent envs only."""
        from .fragment import fragment_words
        return fragment_words(self.U, self.B)

    def _require_compact(self, packed, steps=1):
        if self.kind != _lib.MULTI:
            raise NotImplementedError("compact observation records exist for multi-agent observations (central observations carry no per-env columns)")
        self._require(packed, torch.int32, steps * self.E * self.compact_words, 'packed')

    def step_compact(self, action, packed, reward):
        """step() that writes the observation as the lossless COMPACT record (int32 [E, U (B + 2) + 2B]: per UE dr[B] | utility |
        connection mask, then ues_at_bs[B] | util_at_bs[B] once per env; variants.py:271-305) INSTEAD of the [E, U, 4B + 1] rows: a
        third of the store traffic, and nothing to pack before a learner hand-off.  FragmentCodec(U, B).unpack(packed) is
        bit-identical to the rows step() would have written.  self.obs is NOT updated by this call."""
        self._require(action, torch.uint8, self.E * self.U, 'action')
        self._require_compact(packed)
        self._require(reward, torch.float32, self.reward.numel(), 'reward')
        out = self._make_out(None, reward, packed)
        with torch.cuda.device(self.device):
            self._launch_step(action, out)

    def reset_compact(self, packed):
        """reset() whose first observation is written as the compact record (see step_compact)."""
        self._require_compact(packed)
        return self.reset(_out=self._make_out(None, self.reward, packed))

    def rollout(self, actions, out=None, horizon=None, new_episode_draws=None, _policy_steps=0):
        """T consecutive steps from an action tape [T, E, U] (uint8) in ONE host call -- and, for the narrow kernel
        (``fused_rollout``), ONE kernel launch with the UE state in registers in between (replaces the per-step loop of
        simulation.py:512-541).

        out: None -> the outputs of the last step land in self.obs / self.reward / info tensors; or a dict with 'obs'
        [T, *obs.shape] and 'reward' [T, *reward.shape] (optionally 'sum_utility' [T, E], 'ue_dr' / 'ue_utility' [T, E, U]):
        the outputs of EVERY step (a rollout fragment); multi-agent envs: 'obs_compact' (int32 [T, E, compact_words]) INSTEAD of
        'obs' has every step write the lossless compact record itself (see step_compact).  horizon: reset the envs inside the rollout whenever env.time has
        reached it (RLlib's horizon = episode_length, env_setup.py:281); same sequence as `if time == L: reset()` before
        every step."""
        if actions.dim() != 3:
            raise ValueError("actions must be [T, E, U]")
        T = int(actions.shape[0])
        self._require(actions, torch.uint8, T * self.E * self.U, 'actions')
        if _policy_steps:                        # closed loop (rollout_policy): the tape is one step, the rest is decided in the kernel
            T = int(_policy_steps)
        L = int(horizon or 0)
        if new_episode_draws is None:
            new_episode_draws = self.rand_episodes
        if L and self.rng_mode == _lib.RNG_TAPE and (self.rand_episodes or new_episode_draws or self.dynamic or self._live is not None) and not _policy_steps:
            # rng='reference' with streams that continue across episodes (or UEs re-seeded by list position at reset, or a fixed-episode
            # env whose RUNNING episode was re-seeded with seed(immediate=True): its tape is a splice that only reset() replaces with
            # the configured seed's again, base.py:171-173): every episode
            # needs a tape the HOST draws from where the previous one stopped, so the rollout is cut at the episode boundaries --
            # one launch per stretch, reset() (cursors read back, new tape) in between.  Same sequence as `if time == L: reset()`
            # before every step, like the in-kernel reset of the other modes.
            keys = ('obs', 'obs_compact', 'reward', 'sum_utility', 'ue_dr', 'ue_utility', 'reward_before')
            t0 = 0
            while t0 < T:
                if self.time >= L:
                    self.reset()
                n = min(T - t0, L - self.time)
                self.rollout(actions[t0:t0 + n], out=None if out is None else {k: out[k][t0:t0 + n] for k in keys if out.get(k) is not None})
                t0 += n
            return (self.obs, self.reward) if out is None else (_fragment_obs(out), out['reward'])
        o = self._out
        if out is not None:
            packed = out.get('obs_compact')          # the compact record of every step instead of the rows (see step_compact)
            if packed is not None:
                if out.get('obs') is not None:
                    raise ValueError("out['obs'] and out['obs_compact'] are alternatives")
                self._require_compact(packed, T)
            else:
                self._require(out['obs'], torch.float32, T * self.obs.numel(), "out['obs']")
            self._require(out['reward'], torch.float32, T * self.reward.numel(), "out['reward']")
            ptr = {}
            for k, n in (('sum_utility', self.E), ('ue_dr', self.E * self.U), ('ue_utility', self.E * self.U),
                         ('reward_before', self.E * self.U)):
                if out.get(k) is not None:
                    self._require(out[k], torch.float32, T * n, f"out['{k}']")
                    ptr[k] = out[k].data_ptr()
            o = _lib.DcompOut(None if packed is not None else out['obs'].data_ptr(), out['reward'].data_ptr(), ptr.get('sum_utility'),
                              ptr.get('ue_dr'), ptr.get('ue_utility'), ptr.get('reward_before'),
                              packed.data_ptr() if packed is not None else None)
        if self.rng_mode == _lib.RNG_TAPE:
            self._ensure_tape(min(T, L - self.time) if L else T)
        opts = _lib.DcompRolloutOpts(1 if out is not None else 0, L, 1 if new_episode_draws else 0, 1 if _policy_steps else 0)
        if self.dynamic:
            keep = self._rollout_events(T, L, opts)          # host / device arrays the call reads: alive until it has been enqueued
        with torch.cuda.device(self.device):
            _lib.check(self._L.dcomp_rollout_ex(self._h, self._st_ref, ctypes.c_void_p(actions.data_ptr()), T, ctypes.byref(o),
                                                ctypes.byref(opts), self._stream()))
        if self._policy_key is not None:
            self._policy_launched()
        return (self.obs, self.reward) if out is None else (_fragment_obs(out), out['reward'])

    def _rollout_events(self, T, L, opts):
        """UE departures / arrivals of the next T steps (base.py:433-443) for dcomp_rollout_ex's event feed: the schedule is
        configuration (identical in every env, indexed by env.time, starting over after a reset at the horizon); in
        rng='reference' mode the list positions / border points are drawn here, in step order, from the same host streams
        step() uses.  Returns the arrays to keep alive."""
        n_rem, n_add = np.zeros(T, dtype=np.int32), np.zeros(T, dtype=np.int32)
        rem_blocks, add_blocks = [], []
        t_env, cur = self.time, self.num_ue
        for t in range(T):
            if L and t_env == L:
                t_env, cur = 0, self.U0
            r, a = self.schedule[t_env] if t_env < len(self.schedule) else (0, 0)
            n_rem[t], n_add[t] = r, a
            if self.rng_mode == _lib.RNG_TAPE:
                if r:
                    rem_blocks.append(self._dyn_streams.departures(r, cur).astype(np.int32).reshape(-1))
                if a:
                    add_blocks.append(self._dyn_streams.arrivals(a).astype(np.int32).reshape(-1))
            cur += a - r
            t_env += 1
        keep = [n_rem, n_add]
        opts.ev_n_remove, opts.ev_n_add = n_rem.ctypes.data, n_add.ctypes.data
        for name, blocks in (('ev_remove_idx', rem_blocks), ('ev_add_xy', add_blocks)):
            if blocks:
                d = torch.from_numpy(np.concatenate(blocks)).to(self.device)
                keep.append(d)
                setattr(opts, name, d.data_ptr())
        self._ev_keep = keep
        return keep

    def rollout_policy(self, num_steps, out=None, horizon=None):
        """num_steps steps of the closed loop `act = policy(obs); step(act)` with the policy registered through set_policy().
        ONE call (dcomp_rollout_ex, policy_loop).  On the fused kernel: one launch per stretch of an episode -- the
        decisions never leave the registers -- and at the horizon a reset launch that also decides the first action of the new
        episode; no host work in between.  Where rollouts are not fused (wide / generic kernels, UE arrival / departure): one
        launch per step, enqueued by the library, every step writing straight into its slice of `out`.  Needs a
        current next_action: call it after reset() / step() with the policy set.  out: as in rollout(), [num_steps, ...]
        buffers of every step."""
        if self._policy_key is None or not self._next_action_fresh:
            raise RuntimeError("rollout_policy() needs set_policy() and a reset() / step() after it")
        L = int(horizon or 0)
        T, t0 = int(num_steps), 0
        keys = ('obs', 'obs_compact', 'reward', 'sum_utility', 'ue_dr', 'ue_utility', 'reward_before')
        if out is not None and out.get('obs_compact') is not None:
            self._require_compact(out['obs_compact'], T)
        host_resets = L and self.rng_mode == _lib.RNG_TAPE and (self.rand_episodes or self.dynamic or self._live is not None)   # a fresh host-drawn tape per episode
        if not host_resets:
            return self.rollout(self.next_action.view(1, self.E, self.U), out=out, horizon=L, _policy_steps=T)
        while t0 < T:
            if self.time >= L:
                self.reset()
            n = min(T - t0, L - self.time)
            frag = None if out is None else {k: out[k][t0:t0 + n] for k in keys if out.get(k) is not None}
            self.rollout(self.next_action.view(1, self.E, self.U), out=frag, _policy_steps=n)
            t0 += n
        return (self.obs, self.reward) if out is None else (_fragment_obs(out), out['reward'])

    def heuristic_actions(self, policy, epsilon=0.0, cluster_mask=None, obs=None, out=None):
        """The reference's heuristic baselines (deepcomp/agent/heuristics.py) for every (env, UE) in one launch
        (dcomp_heuristic_actions): reads the packed observation tensor (default: the one the last reset / step wrote) and
        returns the uint8 [E, U] action tensor step() takes.  policy: '3gpp' | 'fullcomp' | 'dynamic' (epsilon) | 'cluster'
        (cluster_mask: int32/uint32 [B] device tensor, bit o of word b = cell o in b's cluster); agents.py wraps this."""
        obs = self.obs if obs is None else obs
        self._require(obs, torch.float32, self.obs.numel(), 'obs')
        if out is None:
            out = torch.empty((self.E, self.U), dtype=torch.uint8, device=self.device)
        self._require(out, torch.uint8, self.E * self.U, 'out')
        if policy not in _lib.POLICY:
            raise ValueError(f"policy must be one of {sorted(_lib.POLICY)}")
        cm = None
        if policy == 'cluster':
            if cluster_mask is None:
                raise ValueError("policy 'cluster' needs cluster_mask")
            self._require(cluster_mask, torch.int32, self.B * (1 if self.B <= _lib.MASK32_MAX_BS else 2), 'cluster_mask')   # > 32 stations: [B, 2] words (lo, hi)
            cm = cluster_mask.data_ptr()
        p = _lib.DcompPolicy(_lib.POLICY[policy], self.kind, self.E, self.U, self.B, self.num_ue, float(epsilon), cm)
        with torch.cuda.device(self.device):
            _lib.check(self._L.dcomp_heuristic_actions(ctypes.byref(p), obs.data_ptr(), out.data_ptr(), self._stream()))
        return out

    def set_policy(self, policy, epsilon=0.0, cluster_mask=None):
        """Heuristic policy INSIDE the step (dcomp_set_policy): from now on every reset / step / rollout launch also writes
        ``self.next_action`` (uint8 [E, U]) = the policy's action on the observation it writes -- what heuristic_actions()
        returns on self.obs, without the second pass over the tensor.  Closed loop: ``env.step(env.next_action)``
        (next_action alternates between two buffers, so the tensor a step was given is intact until the step after).
        policy None switches it off.  Returns False (and leaves it off) if the library refuses (DCOMP_EUNSUPPORTED; no kernel
        does today), where the caller falls back to heuristic_actions(); agents.py::*.act(env) does all of this."""
        if policy is None:
            _lib.check(self._L.dcomp_set_policy(self._h, None, None))
            self._policy_key, self.next_action = None, None
            return True
        if policy not in _lib.POLICY:
            raise ValueError(f"policy must be one of {sorted(_lib.POLICY)}")
        cm = None
        if policy == 'cluster':
            if cluster_mask is None:
                raise ValueError("policy 'cluster' needs cluster_mask")
            self._require(cluster_mask, torch.int32, self.B * (1 if self.B <= _lib.MASK32_MAX_BS else 2), 'cluster_mask')   # > 32 stations: [B, 2] words (lo, hi)
            cm = cluster_mask.data_ptr()
        # two buffers, written alternately: the tensor handed to step() stays what the caller read until the step after
        # (and with UE arrival / departure slots shift, so a step must not write the tensor it reads its actions from)
        bufs = [torch.zeros((self.E, self.U), dtype=torch.uint8, device=self.device) for _ in range(2)]
        p = _lib.DcompPolicy(_lib.POLICY[policy], self.kind, self.E, self.U, self.B, self.num_ue, float(epsilon), cm)
        rc = self._L.dcomp_set_policy(self._h, ctypes.byref(p), bufs[0].data_ptr())
        if rc == _lib.EUNSUPPORTED:
            self._policy_key, self.next_action = None, None
            return False
        _lib.check(rc)
        self._policy_bufs, self._policy_p, self._policy_cm, self._policy_flip = bufs, p, cluster_mask, 0
        self.next_action = bufs[0]
        self._policy_key = (policy, float(epsilon), cm)
        self._next_action_fresh = False              # valid once a reset / step has run with the policy set
        return True

    def _policy_launched(self):
        """Bookkeeping after a launch that wrote next_action; flips the two buffers of a dynamic env for the next launch."""
        self.next_action = self._policy_bufs[self._policy_flip]
        self._next_action_fresh = True
        if len(self._policy_bufs) == 2:
            self._policy_flip ^= 1
            _lib.check(self._L.dcomp_set_policy(self._h, ctypes.byref(self._policy_p), self._policy_bufs[self._policy_flip].data_ptr()))

    @property
    def lanes_per_env(self):
        """Lanes an env occupies in step(): next power of two >= U, or U itself when the batch is packed tightly."""
        return self._L.dcomp_lanes_per_env(self._h)

    @property
    def step_kernel_name(self):
        """The kernel instantiation step() launches, as rocprofv3 prints it ('step_kernel<10, 32, 2>'); None with an older library."""
        if not hasattr(self._L, 'dcomp_step_kernel_name'):
            return None
        buf = ctypes.create_string_buffer(96)
        _lib.check(self._L.dcomp_step_kernel_name(self._h, buf, 96))
        return buf.value.decode()

    @property
    def fused_rollout(self):
        """True when rollout() runs its T steps in one kernel launch."""
        return self._L.dcomp_rollout_is_fused(self._h) == 1

    def rollout_is_fused(self, num_steps, every_step=True, policy_loop=False):
        """True when a rollout of `num_steps` steps is ONE kernel launch (fusion of the short-row shapes needs >= 4 steps or a
        policy loop; fragments of >= 2^31 rows go out one launch per step)."""
        if not hasattr(self._L, 'dcomp_rollout_fused_for'):
            return self.fused_rollout
        return self._L.dcomp_rollout_fused_for(self._h, int(num_steps), 1 if every_step else 0, 1 if policy_loop else 0) == 1

    # ------------------------------------------------------------------ checkpoint / resume
    def _fingerprint(self):
        # seed: the key the running episode draws from (a seed staged by seed() travels as 'pending_seed' next to it)
        return dict(E=self.E, U=self.U, U0=self.U0, B=self.B, kind=int(self.kind), reward=self.reward_agg, seed=int(self._device_seed),
                    env_id_base=self.env_id_base, map=(self.map_w, self.map_h), rand_episodes=self.rand_episodes,
                    bs=self._bs_x.tolist() + self._bs_y.tolist() + self._bs_sh.tolist(), dynamic=self.dynamic,
                    movement=self._pause.tolist() + self._border.tolist(),
                    velocity=self._vlo.tolist() + self._vhi.tolist() + ([] if self._vel_num is None else self._vel_num.tolist()),
                    state_layout=3)

    def state_dict(self):
        """Everything needed to continue this env batch bit-identically in another process (the reference never checkpoints
        env state -- simulation.py:143-147 saves the learner only): the state tensors, the outputs of the last step and the
        handle's five counters.  Counter-based draws only: the stdlib-`random` streams of rng='reference' are host objects."""
        if self.rng_mode != _lib.RNG_PHILOX:
            raise NotImplementedError("state_dict() needs rng='philox' (counter-based draws)")
        c = (ctypes.c_int64 * 5)()
        _lib.check(self._L.dcomp_get_counters(self._h, c))
        torch.cuda.current_stream(self.device).synchronize()
        sd = {'config': self._fingerprint(), 'counters': list(c), 'outbuf': self._outbuf.detach().cpu().clone(),
              'pending_seed': int(self.seed_value) if self._reseeded else None}      # seed() since the last reset(): in force from the next one
        # the registered policy's decision for the NEXT step: saved as it is.  The last launch may have written its observation somewhere
        # else than self.obs (step_into, step_compact, rollout(out=...)), so it cannot be re-derived from `outbuf` on the other side.
        sd['next_action'] = (self.next_action.detach().cpu().clone()
                             if self._policy_key is not None and self.next_action is not None and self._next_action_fresh else None)
        for k in ('pos', 'mv', 'conn', 'conn_hi', 'ewma', 'conn_since', 'uid', 'orig_consumed'):
            t = getattr(self, k)
            sd[k] = None if t is None else t.detach().cpu().clone()
        return sd

    def load_state_dict(self, sd):
        if self.rng_mode != _lib.RNG_PHILOX:
            raise NotImplementedError("load_state_dict() needs rng='philox' (counter-based draws)")
        want, have = self._fingerprint(), dict(sd['config'])
        layout = have.get('state_layout')
        if layout == 2:
            # round-3 checkpoints: no 'velocity' key (fixed non-integer velocities did not exist), 'seed' = the seed of the running episode
            # as well (seed() on a live env was not checkpointable: no pending_seed) -- same tensors, same counters
            have['velocity'], have['state_layout'] = want['velocity'], want['state_layout']
        elif layout != want['state_layout']:
            raise ValueError(f"checkpoint format v{layout} is not supported: this build reads v2 and v{want['state_layout']} "
                             f"(state_dict() writes v{want['state_layout']})")
        if have != want:
            diff = sorted(k for k in set(have) | set(want) if have.get(k) != want.get(k))
            raise ValueError(f"checkpoint belongs to a differently configured env batch (differs in: {', '.join(diff)})")
        for k in ('pos', 'mv', 'conn', 'conn_hi', 'ewma', 'conn_since', 'uid', 'orig_consumed'):
            if sd.get(k) is not None and getattr(self, k) is not None:      # (e.g. conn_hi of a checkpoint written under DCOMP_FORCE_BIG, read without it: stations 32-63 do not exist)
                getattr(self, k).copy_(sd[k])
        self._outbuf.copy_(sd['outbuf'])
        self.flags.zero_()
        c = (ctypes.c_int64 * 5)(*sd['counters'])
        _lib.check(self._L.dcomp_set_counters(self._h, c))
        if sd.get('pending_seed') is not None:          # the checkpointed run had called seed(s) and not yet reset(): so has this one now
            self.seed(sd['pending_seed'])
        elif self._reseeded:                            # a seed staged in THIS env is not part of the checkpointed run
            self.seed_value = self._device_seed
            self.env_seeds = self.seed_value + self._seed_stride * (self.env_id_base + np.arange(self.E, dtype=np.int64))
            self._reseeded = False
        if self._policy_key is not None:
            if sd.get('next_action') is not None:
                # the checkpointed run's own decision for its next step (whatever buffer its last observation went to)
                self.next_action.copy_(sd['next_action'])
            else:
                # a checkpoint written without a registered policy (or before next_action was saved): decide on the restored observation --
                # the stand-alone policy kernel gives what the in-step policy would have written (tests/test_adapters_gpu.py holds the two
                # equal).  Only right if the checkpointed run's last launch wrote self.obs (step / reset / rollout without out=).
                policy, eps, _ = self._policy_key
                self.heuristic_actions(policy, epsilon=eps, cluster_mask=self._policy_cm, obs=self.obs, out=self.next_action)
            self._next_action_fresh = True

    def info(self):
        """base.py:383-411 as tensors."""
        if not self.log_metrics:
            return {'time': self.time}
        m = self.__dict__.get('_metric_views')
        if m is None:                                   # the tensors are fixed views of the output buffer: build the dicts once
            m = self._metric_views = ({'sum_utility': self.sum_utility}, {'dr': self.ue_dr, 'utility': self.ue_utility})
        return {'time': self.time, 'scalar_metrics': m[0], 'vector_metrics': m[1]}

    def enable_reward_before(self):
        """Also write the per-UE pre-move reward clip(utility)/20 (base.py:158-167, 446): the single-agent env's reward."""
        self.want_reward_before = True
        self._out = self._make_out(self.obs, self.reward)
        self._out_ref = ctypes.byref(self._out)

    def outputs_host(self, synced=False, persistent=False, slot=0):
        """Everything the last reset()/step() produced as a dict of numpy views: ONE device->host copy, or (host_io) the
        pinned buffer itself -- then the caller has synchronised (check()) and consumes the views before the next step.
        persistent: the copy lands in ONE pinned host buffer that lives as long as the env, and the SAME dict of views over it is
        returned every time (refilled in place, nothing allocated per call): for callers that build per-env views of it once and
        consume a step's values before the next step (deepcomp_amd.rllib_adapter).  slot: WHICH persistent buffer (each slot is its own
        pinned buffer + view dict): the adapters keep a step's outputs (slot 0) apart from a reset's (slots 1 / 2), so that the views of the
        last step survive a reset that RLlib requests env by env."""
        if self.host_io:
            if not synced:
                torch.cuda.current_stream(self.device).synchronize()
            h = self._outbuf.numpy()
        elif persistent:
            mirrors = self.__dict__.setdefault('_host_mirrors', {})
            if slot not in mirrors:
                m = torch.empty(self._outbuf.shape, dtype=torch.float32).pin_memory()
                hv = m.numpy()
                mirrors[slot] = (m, {name: hv[o:o + cnt].reshape(shape) for name, (o, cnt, shape) in self._sections.items()})
            m, views = mirrors[slot]
            m.copy_(self._outbuf, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
            return views
        else:
            h = self._outbuf.cpu().numpy()
        return {name: h[o:o + cnt].reshape(shape) for name, (o, cnt, shape) in self._sections.items()}

    def obs_views_host(self, host):
        B, U = self.B, self.U
        o = host['obs']
        if self.kind == _lib.MULTI:
            return {'connected': o[..., 0:B], 'dr': o[..., B:2 * B], 'ues_at_bs': o[..., 2 * B:3 * B],
                    'util_at_bs': o[..., 3 * B:4 * B], 'utility': o[..., 4 * B:4 * B + 1]}
        return {'connected': o[..., 0:U * B], 'dr': o[..., U * B:2 * U * B], 'utility': o[..., 2 * U * B:]}

    def check(self):
        """Synchronise and raise what the reference would have asserted (bad action, UE outside the map)."""
        with torch.cuda.device(self.device):
            _lib.check(self._L.dcomp_check(self._h, ctypes.byref(self._st), self._stream()))

    # ------------------------------------------------------------------ views
    def obs_views(self, obs=None):
        """Named slices of the packed observation (keys of variants.py:255-269 / central.py:147-151)."""
        o = self.obs if obs is None else obs
        B, U = self.B, self.U
        if self.kind == _lib.MULTI:
            return {'connected': o[..., 0:B], 'dr': o[..., B:2 * B], 'ues_at_bs': o[..., 2 * B:3 * B],
                    'util_at_bs': o[..., 3 * B:4 * B], 'utility': o[..., 4 * B:4 * B + 1]}
        return {'connected': o[..., 0:U * B], 'dr': o[..., U * B:2 * U * B], 'utility': o[..., 2 * U * B:]}

    def _vel_host(self, vel, uid=None):
        """Velocities as numbers: the movement word's integer, or the configured number of a UE whose velocity is not one
        (uid: the id words per slot of an env whose UE list changes -- arrived UEs are always 'slow')."""
        if self._vel_num is None:
            return vel
        if uid is None:
            fixed = self._vel_num >= 0
            vel[:, fixed] = self._vel_num[fixed]
            return vel
        ids, born = (uid & 0x7FFF).astype(np.int64), (uid & 0x8000) != 0
        num = np.where((ids >= 1) & (ids <= self.U0) & ~born, self._vel_num[np.clip(ids - 1, 0, self.U0 - 1)], -1.0)
        return np.where(num >= 0, num, vel)

    def state_host(self):
        """Host copy of the raw state (parity dumps)."""
        mv = self.mv.cpu().numpy().astype(np.uint64)
        E, U = self.E, self.U
        return {
            'pos': self.pos.cpu().numpy().reshape(E, U, 2),
            'wp': np.stack([(mv & 0xFFFF).astype(np.float64), ((mv >> 16) & 0xFFFF).astype(np.float64)], -1).reshape(E, U, 2),
            'vel': self._vel_host(((mv >> 32) & 0xFF).astype(np.float64).reshape(E, U),
                                  self.uid.cpu().numpy().astype(np.uint16).reshape(E, U) if self.dynamic else None),
            'pausing': ((mv >> 47) & 1).astype(np.int32).reshape(E, U),
            'curr_pause': ((mv >> 40) & 0x7F).astype(np.int32).reshape(E, U),
            'cursor': ((mv >> 48) & 0xFFFF).astype(np.int32).reshape(E, U),
            # one mask per UE: uint32, or uint64 with more than 32 stations (conn | conn_hi << 32)
            'conn': (self.conn.cpu().numpy().astype(np.uint32).reshape(E, U) if self.B <= _lib.MASK32_MAX_BS else
                     (self.conn.cpu().numpy().astype(np.uint32).astype(np.uint64) |
                      (self.conn_hi.cpu().numpy().astype(np.uint32).astype(np.uint64) << np.uint64(32))).reshape(E, U)),
            'ewma': self.ewma.cpu().numpy().reshape(E, U),
            'uid': (self.uid.cpu().numpy().astype(np.uint16).reshape(E, U) & 0x7FFF) if self.dynamic else
                   np.tile(np.arange(1, U + 1, dtype=np.uint16), (E, 1)),
        }


# ====================================================================================================
class _RefSurfaceEnv(_GymEnv):
    """Shared implementation of the reference-named classes (MobileEnv surface, base.py:20-466); a gym.Env like the
    reference's MobileEnv (base.py:20) whenever gym is importable."""
    KIND = None
    metadata = {'render.modes': ['human']}

    def __init__(self, env_config):
        try:                                   # cooperative: gym.Env / RLlib's MultiAgentEnv may (in other versions) have an __init__
            super().__init__()
        except TypeError:                      # pragma: no cover - a base class that insists on arguments
            pass
        self.episode_length = env_config['episode_length']
        self.map = env_config['map']
        self.bs_list = env_config['bs_list']
        self.ue_list = env_config['ue_list']
        self.original_ue_list = list(self.ue_list)
        self.new_ue_interval = env_config.get('new_ue_interval')
        self.ue_arrival = env_config.get('ue_arrival')
        if self.ue_arrival is not None:                                               # base.py:52-56
            self.ue_arrival = {int(k): v for k, v in sorted((int(k), v) for k, v in self.ue_arrival.items())}
            self.new_ue_interval = None
        self.env_seed = env_config['seed']
        self.rand_episodes = env_config['rand_episodes']
        self.log_metrics = env_config.get('log_metrics', True)
        self.dashboard = env_config.get('dashboard', False)
        self.ue_details = env_config.get('ue_details', False)
        self.reward_agg = env_config['reward']
        self.max_ues = env_config.get('max_ues') or _rng.max_num_ue(len(self.ue_list), self.episode_length, self.ue_arrival,
                                                                    self.new_ue_interval)      # base.py:79-83, 191-210
        assert self.max_ues >= self.num_ue                                            # base.py:84
        self.num_envs = int(env_config.get('num_envs', 1))
        self.batched = bool(env_config.get('batched', self.num_envs > 1))
        self.total_utility = 0
        self.obs = None
        self._host = None
        self.core = BatchedMobileEnv(self.map, self.bs_list, self.ue_list, self.KIND, num_envs=self.num_envs,
                                     seed=self.env_seed, episode_length=self.episode_length, reward=self.reward_agg,
                                     rand_episodes=self.rand_episodes, rng=env_config.get('rng', 'reference' if not self.batched else 'philox'),
                                     device=env_config.get('device', 'cuda'), env_id_base=env_config.get('env_id_base', 0),
                                     env_seeds=env_config.get('env_seeds'),
                                     log_metrics=True, ue_arrival=self.ue_arrival, new_ue_interval=self.new_ue_interval,
                                     max_ues=self.max_ues if (self.ue_arrival or self.new_ue_interval) else None,
                                     host_io=not self.batched)
        for i, ue in enumerate(self.ue_list):
            if hasattr(ue, '_env'):
                ue._env, ue._idx = self, i
        for i, bs in enumerate(self.bs_list):
            if hasattr(bs, '_env'):
                bs._env, bs._idx = self, i
        self._view_cache = None
        self._define_spaces()

    # ---- attribute surface used by callers (env_setup.py:291-309, callbacks.py:21, simulation.py:111,499)
    @property
    def num_bs(self):
        return len(self.bs_list)

    @property
    def num_ue(self):
        return len(self.ue_list)

    @property
    def time(self):
        return self.core.time

    @property
    def current_total_utility(self):
        return float(self._host['sum_utility'][0])

    def seed(self, seed=None):
        """base.py:132-143, to the letter in rng='reference' mode: the running episode continues on the new streams at once
        (BatchedMobileEnv.seed, immediate=True); counter-based draws and changing UE lists stage the seed until reset()."""
        self.core.seed(seed, immediate=True)

    def _host_view(self):
        if self._view_cache is None:
            c = self.core
            conn = c.conn[:c.U].cpu().numpy().astype(np.uint32).astype(np.uint64)
            if c.conn_hi is not None:
                conn = conn | (c.conn_hi[:c.U].cpu().numpy().astype(np.uint32).astype(np.uint64) << np.uint64(32))
            self._view_cache = {
                'pos': c.pos[:c.U].cpu().numpy(), 'ewma': c.ewma[:c.U].cpu().numpy().tolist(),
                'curr_dr': self._host['ue_dr'][0].tolist(), 'utility': self._host['ue_utility'][0].tolist(),
                'num_conn': [int(((conn >> np.uint64(b)) & np.uint64(1)).sum()) for b in range(c.B)],
            }
        return self._view_cache

    def done(self):
        return None                                                                   # base.py:371-381

    def get_max_num_ue(self):
        """base.py:191-209: most UEs listed at the same time within an episode."""
        return _rng.max_num_ue(len(self.original_ue_list), self.episode_length, self.ue_arrival, self.new_ue_interval)

    def get_num_diff_ues(self):
        """base.py:211-225: number of DIFFERENT UEs over an episode (env_setup.py:295 sizes the policy map of
        --separate-agent-nns with it): departures do not free an id."""
        if self.ue_arrival is None:
            return self.get_max_num_ue()
        return len(self.original_ue_list) + sum(a for a in self.ue_arrival.values() if a > 0)

    def render(self, mode='human'):
        raise NotImplementedError("rendering is not part of the device path (SURVEY.md section 2: out of scope)")

    def _refresh_ue_list(self):
        """Mirror env 0's UE list (base.py:592-618): original objects for the initial UEs, new config holders for UEs
        that arrived during the episode."""
        if not self.core.dynamic:
            return
        from .entities import RandomWaypoint, User
        words = self.core.uid[:self.core.U].cpu().numpy().astype(np.uint16)[:self.core.num_ue]
        pos = self.core.pos[:self.core.U].cpu().numpy()
        lst = []
        for slot, w in enumerate(words):
            uid, born = int(w & 0x7FFF), bool(w & 0x8000)
            if not born:
                ue = self.original_ue_list[uid - 1]
            else:
                ue = User(str(uid), self.map, float(pos[slot][0]), float(pos[slot][1]), RandomWaypoint(self.map, velocity='slow'))
                ue._env = self
            ue._idx = slot
            lst.append(ue)
        self.ue_list = lst

    # ---- reset / step
    def reset(self):
        self.total_utility = 0
        self._view_cache = None
        obs = self.core.reset()
        if self.batched:
            return obs
        self.ue_list = list(self.original_ue_list)                                    # base.py:177-182
        for i, ue in enumerate(self.ue_list):
            if hasattr(ue, '_idx'):
                ue._idx = i
        self._host = self.core.outputs_host()
        self.obs = self._format_obs(self._host)
        return self.obs

    def step(self, action):
        self._view_cache = None
        if self.batched:
            return self.core.step(action)
        a = self._action_tensor(action)
        obs, reward, _, info = self.core.step(a)
        self.core.check()
        self._refresh_ue_list()
        self._host = self.core.outputs_host(synced=True)                               # pinned views (check() synchronised)
        self.total_utility += float(self._host['sum_utility'][0])
        self.obs = self._format_obs(self._host)
        return self.obs, self._format_reward(self._host), self.done(), self.info()

    def _action_buf(self):
        """Zeroed numpy view of the pinned action buffer the kernel reads (the previous step has completed: step() ends with
        check(), which synchronises)."""
        a = self.core.action_host.numpy()
        a[:] = 0
        return a

    def _info_dict(self):
        """base.py:383-411"""
        if not self.log_metrics:
            return {'time': self.time}
        dr, ut = self._host['ue_dr'][0].tolist(), self._host['ue_utility'][0].tolist()
        return {'time': self.time,
                'scalar_metrics': {'sum_utility': float(self._host['sum_utility'][0])},
                'vector_metrics': {'dr': {f'UE {ue}': dr[i] for i, ue in enumerate(self.ue_list)},
                                   'utility': {f'UE {ue}': ut[i] for i, ue in enumerate(self.ue_list)}}}


class CentralRelNormEnv(_RefSurfaceEnv):
    """Central single-agent env (DeepCoMP): multi_ue/central.py:9-73,143-152."""
    KIND = 'central'

    def _define_spaces(self):
        B, M = self.num_bs, self.max_ues
        self.action_space = spaces.MultiDiscrete([B + 1 for _ in range(M)])            # central.py:28
        self.observation_space = spaces.Dict({                                         # central.py:147-151
            'connected': spaces.MultiBinary(M * B),
            'dr': spaces.Box(low=0, high=1, shape=(M * B,)),
            'utility': spaces.Box(low=-1, high=1, shape=(M,)),
        })

    def _action_tensor(self, action):
        assert self.action_space.contains(action), f"Action {action} does not fit action space {self.action_space}"   # central.py:61
        a = self._action_buf()
        a[0, :self.num_ue] = np.asarray(action, dtype=np.uint8)[:self.num_ue]            # central.py:63: by list position
        return self.core.action_host

    def _format_obs(self, host):
        v = {k: t[0] for k, t in self.core.obs_views_host(host).items()}
        pad = self.max_ues - self.core.U                                               # central.py:46-55 (dead slots are already zero rows)
        out = {'connected': v['connected'].astype(np.int64).tolist() + [0] * (pad * self.num_bs),
               'dr': v['dr'].tolist() + [0] * (pad * self.num_bs),
               'utility': v['utility'].tolist() + [0] * pad}
        return out

    def _format_reward(self, host):
        return float(host['reward'][0])

    def info(self):
        return self._info_dict()


class MultiAgentMobileEnv(_RefSurfaceEnv, _MultiAgentEnv):
    """Multi-agent env (DD-CoMP / D3-CoMP): multi_ue/multi_agent.py:6-107 on top of variants.py:244-305.  Derives from RLlib's
    MultiAgentEnv like the reference (multi_agent.py:6), so `MultiAgentEnv in env_class.__mro__` (env_setup.py:289,
    simulation.py:46) and RLlib's isinstance dispatch see a multi-agent env."""
    KIND = 'multi'

    def _define_spaces(self):
        B = self.num_bs
        self.action_space = spaces.Discrete(B + 1)                                     # variants.py:17
        self.obs_space_dict = {                                                        # variants.py:255-268
            'connected': spaces.MultiBinary(B),
            'dr': spaces.Box(low=0, high=1, shape=(B,)),
            'utility': spaces.Box(low=-1, high=1, shape=(1,)),
            'ues_at_bs': spaces.Box(low=0, high=1, shape=(B,)),
            'util_at_bs': spaces.Box(low=-1, high=1, shape=(B,)),
        }
        self.observation_space = spaces.Dict(self.obs_space_dict)

    def _action_tensor(self, action):
        a = self._action_buf()
        for i, ue in enumerate(self.ue_list):                                          # multi_agent.py:30 (missing ids: no-op)
            if ue.id in action:
                v = int(action[ue.id])
                if not 0 <= v <= self.num_bs:
                    raise IndexError(f"action {v} of UE {ue.id} is outside [0, {self.num_bs}]")
                a[0, i] = v
        return self.core.action_host

    def _format_obs(self, host):
        v = {k: t[0] for k, t in self.core.obs_views_host(host).items()}
        conn = v['connected'].astype(np.int64).tolist()          # whole-array conversions: one C loop each, not one per element
        dr, ut = v['dr'].tolist(), v['utility'].tolist()
        nb, ub = v['ues_at_bs'].tolist(), v['util_at_bs'].tolist()
        return {ue.id: {'connected': conn[i], 'dr': dr[i], 'utility': ut[i], 'ues_at_bs': nb[i], 'util_at_bs': ub[i]}
                for i, ue in enumerate(self.ue_list)}                                  # multi_agent.py:32-37, variants.py:302-303

    def _format_reward(self, host):
        r = host['reward'][0].tolist()
        return {ue.id: r[i] for i, ue in enumerate(self.ue_list)}

    def done(self):
        d = {ue.id: None for ue in self.ue_list}                                       # multi_agent.py:97-102
        d['__all__'] = None
        return d

    def info(self):
        info = self._info_dict()                                                       # multi_agent.py:104-107
        return {ue.id: info for ue in self.ue_list}


class RelNormEnv(_RefSurfaceEnv):
    """Single-agent env ('--agent single', env_setup.py:27-30): ONE UE acts per step, round robin over the UE list
    (base.py:227-245); the observation is the next UE's RelNorm observation (base.py:350-358, variants.py:271-305), the
    reward the acting UE's pre-move reward (base.py:360-369).  All UEs still move and share rates every step."""
    KIND = 'multi'

    def __init__(self, env_config):
        env_config = dict(env_config)
        env_config.setdefault('reward', 'avg')          # MobileEnv itself never reads it (base.py:27-84)
        super().__init__(env_config)
        self.core.enable_reward_before()

    def _define_spaces(self):
        B = self.num_bs
        self.action_space = spaces.Discrete(B + 1)                                     # variants.py:17
        self.obs_space_dict = {'connected': spaces.MultiBinary(B), 'dr': spaces.Box(low=0, high=1, shape=(B,)),
                               'utility': spaces.Box(low=-1, high=1, shape=(1,)),
                               'ues_at_bs': spaces.Box(low=0, high=1, shape=(B,)),
                               'util_at_bs': spaces.Box(low=-1, high=1, shape=(B,))}     # variants.py:255-268
        self.observation_space = spaces.Dict(self.obs_space_dict)

    def _action_tensor(self, action):
        assert self.action_space.contains(action), f"Action {action} does not fit action space {self.action_space}"   # base.py:238
        a = self._action_buf()
        a[0, self.time % self.num_ue] = int(action)                                    # base.py:243-244
        return self.core.action_host

    def _format_obs(self, host):
        v = {k: t[0] for k, t in self.core.obs_views_host(host).items()}
        i = self.time % self.num_ue                                                    # base.py:357
        return {'connected': [int(x) for x in v['connected'][i]], 'dr': [float(x) for x in v['dr'][i]],
                'utility': [float(v['utility'][i][0])], 'ues_at_bs': [float(x) for x in v['ues_at_bs'][i]],
                'util_at_bs': [float(x) for x in v['util_at_bs'][i]]}

    def _format_reward(self, host):
        return float(host['reward_before'][0][(self.time - 1) % self.num_ue])          # base.py:367-369

    def info(self):
        return self._info_dict()


def get_env_class(env_type):
    """env_setup.py:23-37"""
    assert env_type in ('single', 'central', 'multi'), f"Environment type was {env_type} but has to be one of single/central/multi."
    return {'single': RelNormEnv, 'central': CentralRelNormEnv, 'multi': MultiAgentMobileEnv}[env_type]
