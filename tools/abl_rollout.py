import sys, time, torch
sys.path.insert(0, '/root/repo')
from deepcomp_amd import scenarios
from deepcomp_amd.entities import build_from_scenario
from deepcomp_amd.env import BatchedMobileEnv
E, U, B, T, L = 4096, 10, 5, 100, 100
scn = scenarios.grid_map(B, 'mixed').with_ues(num_slow=U)
m, bs, ues = build_from_scenario(scn)
for kind, lm in (('central', True), ('central', False), ('multi', True)):
    env = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=42, episode_length=L, rng='philox', rand_episodes=True, log_metrics=lm)
    tape = torch.randint(0, B + 1, (T, E, U), device='cuda', dtype=torch.uint8)
    zero = torch.zeros_like(tape)
    frag = {'obs': torch.empty((T,) + tuple(env.obs.shape), device='cuda'), 'reward': torch.empty((T,) + tuple(env.reward.shape), device='cuda')}
    env.reset()
    for name, acts, out in (('every-step', tape, frag), ('last-only', tape, None), ('no-actions every-step', zero, frag)):
        for _ in range(5):
            env.rollout(acts, out=out, horizon=L)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 40
        for _ in range(n):
            env.rollout(acts, out=out, horizon=L)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'{kind} log_metrics={lm} {name}: {dt / (n * T) * 1e6:.3f} us/step')
