"""N>1 path on CPU: env-axis sharding and the rollout all-gather with the gloo backend, world_size 2
(the same code runs on RCCL/xGMI with backend 'nccl').  Also: the oracle's Philox draws are keyed by the
GLOBAL env id, so a sharded run equals the unsharded one."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    """Below the kernel's ephemeral range: a bind(0) port can be handed to any outgoing connection before rank 0 binds it."""
    import random
    for _ in range(64):
        port = random.randrange(20000, 32000)
        s = socket.socket()
        try:
            s.bind(('127.0.0.1', port))
            return port
        except OSError:
            continue
        finally:
            s.close()
    raise RuntimeError('no free port')


def _worker(rank, world, port, total_envs, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from deepcomp_amd.sharded import RolloutGather, shard_bounds
        start, count = shard_bounds(total_envs, world)[rank]
        T, U, D = 3, 4, 5
        # equal-size shards for the flat collective
        assert count == total_envs // world
        env_ids = torch.arange(start, start + count, dtype=torch.float32)
        frag = {
            'obs': (env_ids.view(1, count, 1, 1) * 1000 + torch.arange(T).view(T, 1, 1, 1) * 100 +
                    torch.arange(U).view(1, 1, U, 1) * 10 + torch.arange(D).view(1, 1, 1, D)).contiguous(),
            'reward': (env_ids.view(1, count, 1) + torch.arange(T).view(T, 1, 1) * 0.5).expand(T, count, U).contiguous(),
            'action': torch.full((T, count, U), rank, dtype=torch.uint8),
        }
        ok = True
        # 'p2p': the direct all-gather spelled out (one send to / one receive from every peer in one batch) must give the same tensors
        outs = {algo: RolloutGather(algo=algo).all_gather_async(frag).wait() for algo in ('collective', 'p2p')}
        for k in frag:
            ok &= torch.equal(outs['collective'][k], outs['p2p'][k])
        out = outs['p2p']
        for r in range(world):
            s_r, c_r = shard_bounds(total_envs, world)[r]
            ids = torch.arange(s_r, s_r + c_r, dtype=torch.float32)
            want_obs = (ids.view(1, c_r, 1, 1) * 1000 + torch.arange(T).view(T, 1, 1, 1) * 100 +
                        torch.arange(U).view(1, 1, U, 1) * 10 + torch.arange(D).view(1, 1, 1, D))
            ok &= torch.equal(out['obs'][r], want_obs)
            ok &= bool((out['action'][r] == r).all())
            ok &= torch.equal(out['reward'][r][:, :, 0], ids.view(1, c_r) + torch.arange(T).view(T, 1) * 0.5)
        ok &= out['obs'].shape == (world, T, count, U, D)
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok &= float(t.item()) == float(world)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_rollout_all_gather_world2():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, 16, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_rollout_all_gather_world3_p2p():
    """Three ranks: every rank has two peers, so the p2p form posts two sends and two receives per tensor in one batch."""
    world, port = 3, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, 18, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True, 2: True}


def test_rccl_direct_hints_name_knobs_the_installed_library_reads():
    from deepcomp_amd.sharded import rccl_direct_hints
    h = rccl_direct_hints()
    assert int(h['RCCL_DIRECT_ALLGATHER_THRESHOLD']) >= 1 << 31 and h['NCCL_PROTO'] == 'Simple'
    path = '/opt/rocm/lib/librccl.so'
    if os.path.exists(path):
        blob = open(path, 'rb').read()
        for k in h:
            assert k.encode() in blob, f'{k} is not a knob of the installed librccl'


def test_shard_bounds_partition():
    from deepcomp_amd.sharded import shard_bounds
    for total, world in [(262144, 8), (65536, 1), (10, 3), (7, 7), (32768, 8)]:
        b = shard_bounds(total, world)
        assert b[0][0] == 0 and sum(c for _, c in b) == total
        assert all(b[i][0] + b[i][1] == b[i + 1][0] for i in range(world - 1))
        assert max(c for _, c in b) - min(c for _, c in b) <= 1
    assert shard_bounds(262144, 8)[3] == (98304, 32768)     # BASELINE config 4: 32 768 envs per GPU


def test_sharded_equals_unsharded_oracle():
    """Draws are keyed by the global env id: 2 shards of 6 envs == one batch of 12 (what makes the multi-GPU
    path correct by construction; the device kernels use the same key, tests/test_parity_gpu.py checks that)."""
    from deepcomp_amd import scenarios
    from deepcomp_amd.sharded import shard_bounds
    from oracle import oracle as orc
    scn = scenarios.custom_map('mixed').with_ues(num_slow=3, num_fast=1)

    def batch(start, count):
        envs = []
        for e in range(count):
            o = orc.OracleEnv(int(scn.width), int(scn.height), scn.bs_pos, scn.bs_sharing,
                              [s['velocity'] for s in scn.ue_specs], kind=orc.MULTI)
            o.set_philox(99, start + e)
            envs.append(o)
        return orc.OracleBatch(envs, num_threads=2)

    rng = np.random.default_rng(0)
    acts = rng.integers(0, 5, size=(20, 12, 4)).astype(np.uint8)
    full = batch(0, 12)
    full.reset()
    shards = [batch(s, c) for s, c in shard_bounds(12, 2)]
    for sh in shards:
        sh.reset()
    for t in range(20):
        fo, fr, fc, fp = full.step(acts[t])
        for (s, c), sh in zip(shard_bounds(12, 2), shards):
            o, r, cn, p = sh.step(acts[t, s:s + c])
            assert np.array_equal(o, fo[s:s + c]) and np.array_equal(r, fr[s:s + c])
            assert np.array_equal(cn, fc[s:s + c]) and np.array_equal(p, fp[s:s + c])
