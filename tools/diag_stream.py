#!/usr/bin/env python3
"""GPU box diagnostic: what the FIRST use of a second HIP stream, of the RCCL communicator and of RolloutGather does to the step kernels that
follow (us per step over 100 back-to-back launches, printed twice), against a hand-off in steady state.  usage: python tools/diag_stream.py"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcomp_amd import scenarios
from deepcomp_amd.entities import build_from_scenario
from deepcomp_amd.env import BatchedMobileEnv
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
E, U, B = 65536, 32, 10
m, bs, ues = build_from_scenario(scenarios.grid_map(B, 'mixed').with_ues(num_slow=U))
env = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=42, rng='philox', rand_episodes=True, episode_length=100000)
env.reset()
g = torch.Generator(device='cuda').manual_seed(1)
pool = torch.randint(0, B + 1, (4, E, U), generator=g, device='cuda', dtype=torch.uint8)
def timed(n=100):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): env.step(pool[i & 3])
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for _ in range(4): timed(100)
print('baseline            %.2f %.2f us' % (timed(), timed()))
side = torch.cuda.Stream()
x = torch.zeros(1 << 20, device='cuda')
with torch.cuda.stream(side):
    x.add_(1)
torch.cuda.synchronize()
print('after a side stream kernel      %.2f %.2f us' % (timed(), timed()))
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
print('after init_process_group        %.2f %.2f us' % (timed(), timed()))
src = torch.zeros(E * 33, device='cuda'); dst = torch.empty(E * 33, device='cuda')
dist.all_gather_into_tensor(dst, src); torch.cuda.synchronize()
print('after all_gather (main stream)  %.2f %.2f us' % (timed(), timed()))
with torch.cuda.stream(side):
    w = dist.all_gather_into_tensor(dst, src, async_op=True)
torch.cuda.synchronize()
print('after all_gather (side, async, not waited) %.2f %.2f us' % (timed(), timed()))
w.wait(); torch.cuda.synchronize()
print('after work.wait()               %.2f %.2f us' % (timed(), timed()))
del w
print('after del work                  %.2f %.2f us' % (timed(), timed()))
from deepcomp_amd.sharded import RolloutGather
gather = RolloutGather(use_side_stream=True, reuse_buffers=3)
stage = torch.empty((E, U + 1), device='cuda')
def handoff():
    torch.cat((env.reward.view(E, -1), env.sum_utility.view(E, 1)), dim=1, out=stage)
    return gather.all_gather_async({'reward_and_sum_utility': stage})
h = handoff()
print('after RolloutGather hand-off (pending)  %.2f %.2f us' % (timed(), timed()))
h2 = handoff()
print('after a second one (two pending)        %.2f %.2f us' % (timed(), timed()))
h.wait(); h2.wait()
print('after both waited                       %.2f %.2f us' % (timed(), timed()))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(50): env.step(pool[i & 3])
h3 = handoff()
for i in range(50): env.step(pool[i & 3])
b.record(); torch.cuda.synchronize()
print('100 steps with a hand-off in the middle %.2f us per step' % (a.elapsed_time(b) / 100 * 1e3))
print('next                                    %.2f %.2f us' % (timed(), timed()))
dist.barrier(); torch.cuda.synchronize()
print('after barrier                   %.2f %.2f us' % (timed(), timed()))
t = torch.zeros(1, device='cuda'); dist.all_reduce(t); torch.cuda.synchronize()
print('after all_reduce                %.2f %.2f us' % (timed(), timed(300)))
dist.destroy_process_group()
print('after destroy                   %.2f %.2f us' % (timed(), timed()))
