import sys, time, torch
sys.path.insert(0, '/root/repo')
from deepcomp_amd import scenarios
from deepcomp_amd.entities import build_from_scenario
from deepcomp_amd.env import BatchedMobileEnv
E, U, B, L = 65536, 32, 10, 100
scn = scenarios.grid_map(B, 'mixed').with_ues(num_slow=U)
m, bs, ues = build_from_scenario(scn)
env = BatchedMobileEnv(m, bs, ues, 'multi', num_envs=E, seed=42, episode_length=L, rng='philox', rand_episodes=True)
pool = torch.randint(0, B + 1, (16, E, U), device='cuda', dtype=torch.uint8)
env.reset()
import os
if os.environ.get('PRE') == 'fill':
    buf = torch.empty(436 * 1024 * 1024 // 4, device='cuda')
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < float(os.environ.get('PRE_S', '0.1')):
        for _ in range(20):
            buf.fill_(1.0)
        torch.cuda.synchronize()
    print('preconditioned with fill_', flush=True)
def block(n, tag):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for i in range(n):
        env.step(pool[i & 15])
    th = time.perf_counter() - t0
    b.record()
    torch.cuda.synchronize()
    print(f'{tag}: {n} launches  {a.elapsed_time(b) / n * 1e3:.1f} us/launch (events)  host issue {th / n * 1e6:.1f} us/launch', flush=True)
for k in range(12):
    block(25, f'block {k}')
    if k % 4 == 3:
        env.reset()
block(400, 'long')
block(400, 'long')
