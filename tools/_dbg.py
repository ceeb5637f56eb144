import sys, torch
sys.path.insert(0, '.')
from tests.test_rollout_gpu import _make, _state
kind, U, B, E, reward, sharing = ('multi', 3, 3, 700, 'avg', 'resource-fair')
T = 23
g = torch.Generator(device='cuda').manual_seed(11)
acts = torch.randint(0, B + 1, (T, E, U), generator=g, device='cuda', dtype=torch.uint8)
acts[torch.rand((T, E, U), generator=g, device='cuda') < 0.4] = 0
ref = _make(kind, U, B, E, reward, sharing); ref.reset()
want = {k: [] for k in ('obs', 'reward', 'sum_utility', 'ue_dr', 'ue_utility')}
for t in range(T):
    ref.step(acts[t])
    for k in want: want[k].append(getattr(ref, k).clone())
want = {k: torch.stack(v) for k, v in want.items()}
b = _make(kind, U, B, E, reward, sharing); b.reset()
out = {k: torch.full_like(v, float('nan')) for k, v in want.items()}
b.rollout(acts, out=out)
print(ref.step_kernel_name, b.rollout_is_fused(T))
for k in want:
    d = (out[k] != want[k])
    print(k, int(d.sum()), 'of', d.numel())
    if d.any():
        idx = d.nonzero()[:6]
        for i in idx:
            i = tuple(i.tolist()); print('   ', i, float(out[k][i]), float(want[k][i]), float(out[k][i]) - float(want[k][i]))
o, w = out['obs'], want['obs']
d = (o != w).nonzero()
print('obs columns that differ:', sorted(set(d[:, -1].tolist()))[:20])
