#!/usr/bin/env python3
"""A/B of library builds on ONE GPU box (boxes differ by +-3 %, so only same-box comparisons count).

  python tools/ab/ab_lib.py build <tag> [--ref GITREF] [--b 5,10,32] [--flags "-DX=1 ..."]     (build container)
        -> deepcomp_amd/csrc/variants/libdcomp_hip_<tag>.so from the working tree, or from the sources at GITREF
           (csrc/ + include/ exported to a scratch directory); only the listed base-station counts (seconds, not minutes)
  python tools/ab/ab_lib.py run <tag> <tag> ... [--rounds 2] [--only c3,c2roll,...]              (GPU box, via gpurun)
        -> every workload timed with every library, interleaved `rounds` times, one child process per (library, round);
           prints kernel ms per step (HIP events, steady state) and the ratio to the first tag
  python tools/ab/ab_lib.py measure [--only ...]     (child: the library is whatever DCOMP_LIB names)

Workloads: BASELINE config 3 (65 536 x 32 x 10 multi, one launch per step), config 2 through the fused rollout (4 096 x 10 x 5
central, 100 steps per launch, every step's outputs), one GPU's share of config 5 (4 096 x 128 x 32) and of config 4
(32 768 x 32 x 10), 65 536 x 10 x 5 central, 65 536 x 32 x 10 central, the closed policy loop at config 2."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))     # tools/ab/ -> the repo
CSRC = os.path.join(REPO, 'deepcomp_amd', 'csrc')
VAR = os.path.join(CSRC, 'variants')
BASE = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast-honor-pragmas']


def build(a):
    os.makedirs(VAR, exist_ok=True)
    src = a.src or REPO
    tmp = None
    if a.ref:
        tmp = tempfile.mkdtemp(prefix='dcomp_ref_')
        subprocess.check_call(f'git -C {REPO} archive {a.ref} deepcomp_amd/csrc include | tar -x -C {tmp}', shell=True)
        src = tmp
    bl = [int(x) for x in a.b.split(',')]
    flags = BASE + [f for f in a.flags.split() if f] + ['-DDCOMP_B_LIST(X)=' + ' '.join(f'X({b})' for b in bl),
                                                         '-DDCOMP_B_LIST_STR="' + ','.join(map(str, bl)) + ' (A/B build)"']
    csrc = os.path.join(src, 'deepcomp_amd', 'csrc')
    objdir = tempfile.mkdtemp(prefix='dcomp_obj_')
    procs = []
    for b in bl:
        o = os.path.join(objdir, f'b{b}.o')
        procs.append((o, subprocess.Popen(['hipcc'] + flags + [f'-DDCOMP_B={b}', '-c', os.path.join(csrc, 'dcomp_inst.hip'), '-o', o],
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for name in ('api', 'big'):                  # the ABI + dispatch, and the generic kernel dcomp_api.hip links against (round 5 on)
        if not os.path.exists(os.path.join(csrc, f'dcomp_{name}.hip')):
            continue
        o = os.path.join(objdir, f'{name}.o')
        procs.append((o, subprocess.Popen(['hipcc'] + flags + ['-c', os.path.join(csrc, f'dcomp_{name}.hip'), '-o', o], stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT, text=True)))
    for o, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            sys.exit(out[-4000:])
    so = os.path.join(VAR, f'libdcomp_hip_{a.tag}.so')
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', so] + [o for o, _ in procs] + ['-lpthread'])
    shutil.rmtree(objdir)
    if tmp:
        shutil.rmtree(tmp)
    print(so, os.path.getsize(so) >> 20, 'MiB')


WORKLOADS = ['big64', 'big40c', 'c3', 'c2roll', 'c5', 'c5mid', 'c5big', 'c4share', 'c4big', 'central10x5', 'central10x5roll', 'central10x5f', 'central32x10', 'c2policy', 'c3rf', 'c3roll']


def measure(a):
    sys.path.insert(0, REPO)
    import torch
    import bench
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    dev = torch.device('cuda', 0)
    mk = (torch, BatchedMobileEnv, scenarios, build_from_scenario, dev)
    only = a.only.split(',') if a.only else WORKLOADS
    out = {}
    for w in only:
        if w == 'big64':                               # the generic kernel (dcomp_big.h): 8 192 x 32 x 64 multi / 65 536 x 10 x 40 central
            out[w] = bench.measure_steps(*mk, 8192, 32, 64, 'multi')['kernel_ms']
        elif w == 'big40c':
            out[w] = bench.measure_steps(*mk, 65536, 10, 40, 'central')['kernel_ms']
        elif w == 'c3':
            out[w] = bench.measure_steps(*mk, 65536, 32, 10, 'multi')['kernel_ms']
        elif w == 'c3rf':
            out[w] = bench.measure_steps(*mk, 65536, 32, 10, 'multi', sharing='resource-fair')['kernel_ms']
        elif w == 'c5':
            out[w] = bench.measure_steps(*mk, 4096, 128, 32, 'multi')['kernel_ms']
        elif w == 'c5big':
            out[w] = bench.measure_steps(*mk, 32768, 128, 32, 'multi', steps=100)['kernel_ms']
        elif w == 'c5mid':
            out[w] = bench.measure_steps(*mk, 8192, 128, 32, 'multi', steps=200)['kernel_ms']
        elif w == 'c4big':
            out[w] = bench.measure_steps(*mk, 262144, 32, 10, 'multi', steps=200)['kernel_ms']
        elif w == 'c4share':
            out[w] = bench.measure_steps(*mk, 32768, 32, 10, 'multi')['kernel_ms']
        elif w == 'central10x5':
            out[w] = bench.measure_steps(*mk, 65536, 10, 5, 'central')['kernel_ms']
        elif w == 'central32x10':
            out[w] = bench.measure_steps(*mk, 65536, 32, 10, 'central')['kernel_ms']
        elif w in ('central10x5f', 'c3roll'):      # the fused rollout kernel forced onto a big batch, ONE step per launch
            os.environ['DCOMP_FUSE_MAX_WAVES'] = '100000000'
            E, U, B, kind = (65536, 10, 5, 'central') if w == 'central10x5f' else (65536, 32, 10, 'multi')
            scn = scenarios.grid_map(B, 'mixed').with_ues(num_slow=U)
            m, bs, ues = build_from_scenario(scn)
            env = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=42, episode_length=100, rng='philox', rand_episodes=True, device=dev)
            del os.environ['DCOMP_FUSE_MAX_WAVES']
            assert env.fused_rollout
            g = torch.Generator(device=dev).manual_seed(7)
            pool = torch.randint(0, B + 1, (4, 1, E, U), generator=g, device=dev, dtype=torch.uint8)
            env.reset()
            for i in range(300):
                env.rollout(pool[i & 3])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(300):
                env.rollout(pool[i & 3])
            e1.record()
            torch.cuda.synchronize()
            env.check()
            out[w] = e0.elapsed_time(e1) / 300
        elif w == 'central10x5roll':     # the 65 536-env central batch through rollout(T = 50): fused at any batch size
            out[w] = bench.measure_rollout(*mk, 65536, 10, 5, 'central', T=50, steps=1500)['ms_per_step']
        elif w == 'c2roll':
            out[w] = bench.measure_rollout(*mk, 4096, 10, 5, 'central', T=100, steps=6000)['ms_per_step']
        elif w == 'c2policy':
            scn = scenarios.grid_map(5, 'mixed').with_ues(num_slow=10)
            m, bs, ues = build_from_scenario(scn)
            env = BatchedMobileEnv(m, bs, ues, 'central', num_envs=4096, seed=42, episode_length=100, rng='philox', rand_episodes=True, device=dev)
            env.set_policy('3gpp')
            env.reset()
            env.rollout_policy(300, horizon=100)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            env.rollout_policy(3000, horizon=100)
            e1.record()
            torch.cuda.synchronize()
            env.check()
            out[w] = e0.elapsed_time(e1) / 3000
    print('AB_RESULT ' + json.dumps(out), flush=True)


def run(a):
    res = {t: {} for t in a.tags}
    for r in range(a.rounds):
        for t in a.tags:
            lib = os.path.join(VAR, f'libdcomp_hip_{t}.so') if t != 'tree' else os.path.join(CSRC, 'libdcomp_hip.so')
            env = dict(os.environ, DCOMP_LIB=lib, DCOMP_AB_NO_BUILD='1')
            p = subprocess.run([sys.executable, os.path.abspath(__file__), 'measure'] + (['--only', a.only] if a.only else []), env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith('AB_RESULT ')]
            if not line:
                print(t, 'FAILED', p.stdout[-1500:], flush=True)
                continue
            for k, v in json.loads(line[-1][10:]).items():
                res[t].setdefault(k, []).append(v)
            print(f'round {r} {t}: ' + '  '.join(f'{k} {v * 1e3:.2f}us' for k, v in json.loads(line[-1][10:]).items()), flush=True)
    ref = a.tags[0]
    print(f'\n{"workload":14s}' + ''.join(f'{t:>22s}' for t in a.tags))
    for w in WORKLOADS:
        if w not in res[ref]:
            continue
        base = min(res[ref][w])
        row = f'{w:14s}'
        for t in a.tags:
            if w in res[t]:
                v = min(res[t][w])
                row += f'{v * 1e3:12.2f} us {v / base:6.3f}x'
            else:
                row += f'{"-":>22s}'
        print(row)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest='cmd', required=True)
    b = sub.add_parser('build'); b.add_argument('tag'); b.add_argument('--ref', default=''); b.add_argument('--b', default='5,10,32'); b.add_argument('--flags', default=''); b.add_argument('--src', default='', help='a copy of the tree (with deepcomp_amd/csrc and include) to build instead of the working tree: experiments that leave the product sources alone')
    r = sub.add_parser('run'); r.add_argument('tags', nargs='+'); r.add_argument('--rounds', type=int, default=2); r.add_argument('--only', default='')
    m = sub.add_parser('measure'); m.add_argument('--only', default='')
    a = ap.parse_args()
    {'build': build, 'run': run, 'measure': measure}[a.cmd](a)
