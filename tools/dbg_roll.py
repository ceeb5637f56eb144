import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_rollout_gpu import _make, _state
kind, U, B, E, reward, sharing = ('multi', 3, 3, 700, 'avg', 'resource-fair')
g = torch.Generator(device='cuda').manual_seed(11)
T = 23
acts = torch.randint(0, B + 1, (T, E, U), generator=g, device='cuda', dtype=torch.uint8)
acts[torch.rand((T, E, U), generator=g, device='cuda') < 0.4] = 0
for Tn in (1, 2, 3, 5, 8, 9, 23):
    ref = _make(kind, U, B, E, reward, sharing); ref.reset()
    for t in range(Tn):
        ref.step(acts[t])
    a = _make(kind, U, B, E, reward, sharing); a.reset()
    a.rollout(acts[:Tn].contiguous())
    sa, sr = _state(a), _state(ref)
    for k in sa:
        if not torch.equal(sa[k], sr[k]):
            d = (sa[k] != sr[k])
            if d.dim() > 1: d = d.any(dim=1)
            idx = d.nonzero().flatten()
            print(f'T={Tn} {k}: {len(idx)} rows differ, first {idx[:8].tolist()} ; rollout {sa[k][idx[0]].tolist()} vs steps {sr[k][idx[0]].tolist()}')
    print(f'T={Tn} done, obs equal: {torch.equal(a.obs, ref.obs)}', flush=True)
