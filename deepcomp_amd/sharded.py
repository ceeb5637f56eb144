"""Env-axis sharding over the GPUs of one node and the rollout hand-off to the learner.

The reference scales out with N independent Ray rollout workers, each holding its own env copy, and ships
sample batches to the trainer through Ray's object store (env_setup.py:266, simulation.py:143).  Here env
instances are rows of one batch: rank g owns the contiguous slice [g*E/N, (g+1)*E/N) of the global env axis, the
BS table is replicated (tiny), and RNG draws are keyed by the *global* env id so results do not depend on N.
There is no collective inside ``step``.  The only exchange is handing a rollout fragment (T steps of
observations / rewards / actions) to the learner: ONE all-gather per fragment -- RCCL over xGMI when the tensors
live on GPUs (torch backend "nccl"), gloo in the CPU tests -- issued asynchronously so it overlaps with the next
fragment's stepping.
"""
from dataclasses import dataclass

import torch
import torch.distributed as dist


def shard_bounds(total_envs, world_size):
    """Contiguous partition of the env axis: list of (start, count) per rank; counts differ by at most one."""
    assert total_envs >= world_size >= 1
    base, rem = divmod(total_envs, world_size)
    out, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < rem else 0)
        out.append((start, n))
        start += n
    return out


def make_sharded_env(scenario, kind, total_envs, rank, world_size, device=None, **kw):
    """This rank's BatchedMobileEnv over its slice of the global env axis (``env_id_base`` = slice start)."""
    from .entities import build_from_scenario
    from .env import BatchedMobileEnv
    start, count = shard_bounds(total_envs, world_size)[rank]
    m, bs, ues = build_from_scenario(scenario)
    dev = device if device is not None else torch.device('cuda', rank % max(torch.cuda.device_count(), 1))
    return BatchedMobileEnv(m, bs, ues, kind, num_envs=count, env_id_base=start, device=dev, **kw)


@dataclass
class GatherHandle:
    out: dict
    works: list
    stream: object = None
    codec: object = None            # FragmentCodec: out['obs'] arrived as the compact record and is unpacked by wait()
    unpack_out: object = None       # where the unpacked rows go (a reused buffer of the RolloutGather; None: a fresh tensor)
    check_lossless: bool = True
    error: object = None            # the message of a failed lossless check (every later wait() raises it again)

    def wait(self):
        """Block until the gathered fragment is usable; returns {name: tensor[world, ...fragment shape]}.  A fragment that
        crossed the links as the compact record (RolloutGather(codec=...)) comes back as the ROWS, bit-identical to what every
        rank's step kernel wrote (dcomp_unpack_fragment on the caller's current stream); the record itself stays available as
        out['obs_compact'].  "Bit-identical" is CHECKED, not assumed: every rank's pack kernel raises a flag word when its input was
        not a multi-agent observation tensor (per-env columns that differ between the rows of an env, `connected` entries that are not
        0 / 1), the words travel with the fragment (out['pack_flags'], one per rank) and a set word raises ValueError here -- one host
        synchronisation per hand-off; RolloutGather(check_lossless=False) leaves the words to the caller."""
        if self.error is not None:                # a failed check stays failed: a second wait() must not hand out the lossy rows
            raise ValueError(self.error)
        for w in self.works:
            w.wait()
        self.works = []
        if self.stream is not None:
            torch.cuda.current_stream(self.stream.device).wait_stream(self.stream)
            self.stream = None
        if self.codec is not None and 'obs' not in self.out:
            if self.check_lossless and 'pack_flags' in self.out:        # BEFORE the rows are published
                fl = self.out['pack_flags'].view(-1).tolist()
                if any(fl):
                    why = {1: 'per-env columns differ between the rows of an env', 2: '`connected` entry that is not 0 / 1'}
                    self.error = ('RolloutGather: the compact record is not lossless for this fragment -- ' + '; '.join(
                        f"rank {r}: " + ', '.join(m for b, m in why.items() if f & b) for r, f in enumerate(fl) if f))
                    raise ValueError(self.error)
            self.out['obs'] = self.codec.unpack(self.out['obs_compact'], out=self.unpack_out)
        return self.out


class RolloutGather:
    """All-gathers rollout fragments along a new leading rank axis.

    ``fragment`` is a dict of equally-shaped-per-rank tensors, e.g. ``{'obs': [T, E_local, U, 4B+1],
    'reward': [T, E_local, U], 'action': [T, E_local, U]}``.  Global env id of ``out[name][r, t, e]`` is
    ``shard_bounds(...)[r][0] + e``.  One flat collective per tensor (large messages: on the fully connected
    xGMI mesh every GPU pushes its shard to its 7 peers concurrently)."""

    def __init__(self, group=None, use_side_stream=True, reuse_buffers=0, codec=None, check_lossless=True, algo='collective'):
        """algo: 'collective' = one all_gather_into_tensor per tensor (RCCL picks ring / direct by its own thresholds; rccl_direct_hints()
        moves them); 'p2p' = the direct all-gather spelled out: every rank posts one send of its shard to, and one receive from, each of
        its N - 1 peers in ONE batch (dist.batch_isend_irecv -> one RCCL group call) -- on the fully connected xGMI mesh every pair has
        its own link, so all N - 1 transfers of a rank run concurrently and none is forwarded ("direct (one-shot, all-peers)
        all-gather rather than ring"); the own shard is a device-local copy.  Same result tensor, same handle; gloo runs both.
        reuse_buffers = k > 0: the gathered tensors -- and, with a codec, the unpacked rows wait() returns, the largest tensor of a
        hand-off -- come from k alternating sets of buffers instead of fresh allocations (a hand-off every few steps should not pay
        the allocator: a GB-sized hipMalloc was seen to cost 30 ms): the tensors of a handle are valid until k further calls.
        The side stream, the communicator's channels and the buffer sets come into being with the first k hand-offs, and the first use
        of a second stream is followed by a cold-start-like transient of the step kernels (~200 launches 10-20 % slow,
        tools/diag_stream.py): a sampler that measures itself should hand over k fragments before it starts the clock."""
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.use_side_stream = use_side_stream
        self._stream = None
        self._reuse, self._sets, self._turn = int(reuse_buffers), {}, 0
        # codec (deepcomp_amd.fragment.FragmentCodec): fragment['obs'] (multi-agent rows [..., U, 4B+1]) crosses the links as the
        # lossless compact record -- 3.2-3.7x fewer bytes -- packed here, unpacked by GatherHandle.wait()
        self.codec = codec
        self.check_lossless = bool(check_lossless)      # wait() reads the gathered pack flags (one host sync per hand-off) and raises if set
        if algo not in ('collective', 'p2p'):
            raise ValueError("RolloutGather: algo must be 'collective' or 'p2p'")
        self.algo = algo

    def all_gather_async(self, fragment):
        out, works = {}, []
        if self.codec is not None and 'obs' in fragment:
            fragment = dict(fragment)
            fragment['obs_compact'] = self.codec.pack(fragment.pop('obs').contiguous())      # on the caller's stream, before the hand-over
            # THIS fragment's flag word travels with it (the codec's word is sticky over all of its pack() calls: snapshot, then clear)
            # (so FragmentCodec.check() no longer sees a bad pack made through this class: the travelling word is the signal --
            # GatherHandle.wait() raises on it, or, with check_lossless=False, the caller reads out['pack_flags'])
            fragment['pack_flags'] = self.codec.flags.clone()
            self.codec.flags.zero_()
        some = next(iter(fragment.values()))
        stream = None
        if some.is_cuda and self.use_side_stream:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=some.device)
            stream = self._stream
            stream.wait_stream(torch.cuda.current_stream(some.device))    # fragment fully written before it is sent
        ctx = torch.cuda.stream(stream) if stream is not None else _NullCtx()
        with ctx:
            for name, t in fragment.items():
                t = t.contiguous()
                if self._reuse:
                    key = (name, tuple(t.shape), t.dtype, t.device, self._turn % self._reuse)
                    o = self._sets.get(key)
                    if o is None:
                        o = self._sets[key] = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
                else:
                    o = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
                if self.algo == 'p2p' and self.world > 1:
                    o[self.rank].copy_(t, non_blocking=True)
                    ops = []
                    for d in range(1, self.world):      # rank r sends to r + d while it receives from r - d: every link busy in both directions
                        to, frm = (self.rank + d) % self.world, (self.rank - d) % self.world
                        ops.append(dist.P2POp(dist.isend, t, dist.get_global_rank(self.group, to) if self.group is not None else to, self.group))
                        ops.append(dist.P2POp(dist.irecv, o[frm], dist.get_global_rank(self.group, frm) if self.group is not None else frm, self.group))
                    works.extend(dist.batch_isend_irecv(ops))
                else:
                    works.append(dist.all_gather_into_tensor(o.view(-1), t.view(-1), group=self.group, async_op=True))
                out[name] = o
        unpack_out = None
        if self.codec is not None and 'obs_compact' in out and self._reuse:
            oc = out['obs_compact']
            key = ('obs/unpacked', tuple(oc.shape), oc.device, self._turn % self._reuse)
            unpack_out = self._sets.get(key)
            if unpack_out is None:
                unpack_out = self._sets[key] = torch.empty(tuple(oc.shape[:-1]) + (self.codec.U, 4 * self.codec.B + 1), dtype=torch.float32, device=oc.device)
        self._turn += 1
        return GatherHandle(out, works, stream, self.codec if 'obs_compact' in out else None, unpack_out, self.check_lossless)

    def all_gather(self, fragment):
        return self.all_gather_async(fragment).wait()


def rccl_direct_hints(threshold_bytes=1 << 34):
    """Environment hints that make RCCL take its DIRECT all-gather (each rank writes its shard straight to every peer over that pair's
    own xGMI link) for messages up to threshold_bytes instead of the ring, and the Simple protocol for them.  Must be in the environment
    before the communicator is created (bench.py --rccl-direct sets them before init_process_group); what the installed librccl reads:
    `strings librccl.so` lists RCCL_DIRECT_ALLGATHER_THRESHOLD and NCCL_PROTO.  Returns the dict (the caller updates os.environ)."""
    return {'RCCL_DIRECT_ALLGATHER_THRESHOLD': str(int(threshold_bytes)), 'NCCL_PROTO': 'Simple'}


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class RolloutBuffer:
    """Pre-allocated fragment storage [T, E_local, ...] that ``BatchedMobileEnv.step_into`` writes in place."""

    def __init__(self, env, horizon):
        self.env, self.T = env, horizon
        dev = env.device
        self.obs = torch.zeros((horizon,) + tuple(env.obs.shape), dtype=torch.float32, device=dev)
        self.reward = torch.zeros((horizon,) + tuple(env.reward.shape), dtype=torch.float32, device=dev)
        self.action = torch.zeros((horizon, env.E, env.U), dtype=torch.uint8, device=dev)

    def collect(self, policy):
        """T steps: ``policy(obs_tensor) -> uint8 actions [E, U]`` runs on the device; nothing per env on the host."""
        obs = self.env.obs
        for t in range(self.T):
            a = policy(obs)
            self.action[t].copy_(a)
            self.env.step_into(self.action[t], self.obs[t], self.reward[t])
            obs = self.obs[t]
        return {'obs': self.obs, 'reward': self.reward, 'action': self.action}
