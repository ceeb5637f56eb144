import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
mode = sys.argv[1]
full = os.sched_getaffinity(0)
node0 = set(range(0, 64)) | set(range(128, 192))
if mode in ('pin', 'pin_restore', 'torch_pin'):
    if mode == 'torch_pin':
        import torch
        torch.zeros(10).sum()
    os.sched_setaffinity(0, node0 & full)
if mode == 'torch_nopin':
    import torch
    torch.zeros(10).sum()
import bench
from deepcomp_amd import scenarios
scn = scenarios.grid_map(10, 'mixed').with_ues(num_slow=32)
if mode == 'pin_restore':
    os.sched_setaffinity(0, full)
t = time.time()
r = bench.cpu_baseline(scn, 'multi', 32, 10, budget_s=4)
print(mode, 'cpus', len(os.sched_getaffinity(0)), 'all-cores', round(r['value']), 'threads', r['cores'], 'single', round(r['single_core']['value']), r['sample'], {k: v for k, v in os.environ.items() if 'OMP' in k}, round(time.time() - t, 1), flush=True)
