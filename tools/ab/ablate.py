#!/usr/bin/env python3
"""Timing-only ablation variants of the step kernel (results are WRONG by construction; never shipped).

  python tools/ab/ablate.py build            # here: builds deepcomp_amd/csrc/variants/libdcomp_hip_abl<N>.so for B=10
  python tools/ab/ablate.py run              # on the GPU box: bench each variant, print kernel_ms table
bits: 1 pre-move rates, 2 move, 4 post-move rates, 8 obs stores, 16 per-BS utility sums, 32 pre-move pairs,
      64 post-move pairs, 128 philox
"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))     # tools/ab/ -> the repo
CSRC = os.path.join(REPO, 'deepcomp_amd', 'csrc')
VAR = os.path.join(CSRC, 'variants')
MASKS = [0, 1, 2, 4, 8, 16, 32, 64, 128, 1 | 4, 32 | 64, 8 | 16, 255]
if os.environ.get('ABL_MASKS'):
    MASKS = [int(x) for x in os.environ['ABL_MASKS'].split(',')]
EXTRA = os.environ.get('ABL_FLAGS', '').split()
TAG = os.environ.get('ABL_TAG', '')
ABL_B = os.environ.get('ABL_B', '10')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast-honor-pragmas', f'-DDCOMP_B_LIST(X)=X({ABL_B})',
         f'-DDCOMP_B_LIST_STR="{ABL_B}"']


def build():
    os.makedirs(VAR, exist_ok=True)
    procs = []
    for m in MASKS:
        so = os.path.join(VAR, f'libdcomp_hip_abl{m}{TAG}.so')
        cmd = ['hipcc'] + FLAGS + EXTRA + [f'-DDCOMP_ABLATE={m}', f'-DDCOMP_B={ABL_B}', '-shared', os.path.join(CSRC, 'dcomp_inst.hip'),
                                   os.path.join(CSRC, 'dcomp_api.hip'), os.path.join(CSRC, 'dcomp_big.hip'), '-o', so, '-lpthread']
        procs.append((m, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        if len(procs) % 6 == 0:
            for _, p in procs[-6:]:
                p.wait()
    for m, p in procs:
        out, _ = p.communicate()
        print(m, 'ok' if p.returncode == 0 else out[-2000:])


def run():
    rows = []
    for m in MASKS:
        env = dict(os.environ, DCOMP_LIB=os.path.join(VAR, f'libdcomp_hip_abl{m}{TAG}.so'))
        r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--steps', '200', '--warmup', '20', '--no-cpu-baseline', '--no-also', '--no-stream',
                            '--no-check'] + sys.argv[2:], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        if not line:
            print(m, 'FAILED', r.stdout[-500:])
            continue
        j = json.loads(line[-1])
        rows.append((m, j['roofline']['kernel_ms'], j['ms_per_step']))
        print(f'{TAG} ablate={m:3d}  kernel_ms={j["roofline"]["kernel_ms"]:.4f}  ms_per_step={j["ms_per_step"]:.4f}', flush=True)


if __name__ == '__main__':
    {'build': build, 'run': run}[sys.argv[1]]()
