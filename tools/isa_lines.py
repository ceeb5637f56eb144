#!/usr/bin/env python3
"""Static gfx950 ISA of ONE kernel attributed to source lines / functions (no GPU needed) -- how round 5 found ~120 removable vector instructions
per wave-step in the headline kernel once the clock trace had shown it on its VALU issue floor (DESIGN_LOG R5.6).

    python tools/isa_lines.py --b 10 --upad 32 [--kernel 'step_kernelILi10ELi32ELi2E'] [--top 40] [--function write_outputs]

Compiles deepcomp_amd/csrc/dcomp_inst.hip for one station count / lane width with line tables (hipcc -S -gline-tables-only --cuda-device-only, ~25 s),
takes the named kernel's body and counts instructions by class per source function (the innermost inlined function of each .loc) and per source line.
Static counts: rare branches are in there, loops count once.  tools/isa_stats.py has the per-kernel totals and register counts."""
import argparse
import bisect
import collections
import os
import re
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, 'deepcomp_amd', 'csrc')


def classify(op):
    if op.startswith(('v_log', 'v_exp', 'v_rcp', 'v_rsq', 'v_sqrt')):
        return 'trans'
    if op.startswith('v_') and 'f64' in op:
        return 'valu64'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    return 'other'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--b', type=int, default=10)
    ap.add_argument('--upad', type=int, default=32)
    ap.add_argument('--kernel', default=None, help='substring of the mangled kernel name (default: step_kernel<B, UPAD, 2>)')
    ap.add_argument('--top', type=int, default=30)
    ap.add_argument('--function', default=None, help='also list every line of this source function')
    ap.add_argument('--flags', default='')
    ap.add_argument('--keep', default='/tmp/isa_lines')
    a = ap.parse_args()
    os.makedirs(a.keep, exist_ok=True)
    asm = os.path.join(a.keep, f'b{a.b}_u{a.upad}.s')
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast-honor-pragmas', f'-DDCOMP_B={a.b}', f'-DDCOMP_ONLY_UPAD={a.upad}',
           '-gline-tables-only', '-S', '--cuda-device-only', '-o', asm, os.path.join(CSRC, 'dcomp_inst.hip')] + [f for f in a.flags.split() if f]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    want = a.kernel or f'step_kernelILi{a.b}ELi{a.upad}ELi2E'
    L = open(asm).read().split('\n')
    starts = [i for i, l in enumerate(L) if re.match(r'^_Z\w+:', l) and want in l]
    if not starts:
        raise SystemExit(f'no kernel matching {want!r} in {asm}')
    start = starts[0]
    end = next(i for i in range(start, len(L)) if L[i].strip().startswith('s_endpgm'))
    files = {}
    for l in L:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
    src = {f: open(os.path.join(CSRC, f)).read().split('\n') for f in ('dcomp_device.h', 'dcomp_wide.h', 'dcomp_dyn.h', 'dcomp_big.h') if os.path.exists(os.path.join(CSRC, f))}
    funcs = {}
    for f, lines in src.items():
        funcs[f] = [(i, m.group(1)) for i, l in enumerate(lines, 1) for m in [re.match(r'^(?:__device__|__global__).*?\b(\w+)\s*\(', l)] if m]
    cur = ('?', 0)
    per_line = collections.defaultdict(collections.Counter)
    for i in range(start, end + 1):
        l = L[i]
        m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
        if m:
            cur = (files.get(int(m.group(1)), '?'), int(m.group(2)))
            continue
        t = l.strip()
        if not t or t.startswith(('.', ';', '//')) or t.endswith(':'):
            continue
        per_line[cur][classify(t.split()[0])] += 1

    def func_of(f, ln):
        fl = funcs.get(f)
        if not fl:
            return f
        k = bisect.bisect_right([x[0] for x in fl], ln) - 1
        return fl[k][1] if k >= 0 else f
    per_func = collections.defaultdict(collections.Counter)
    for (f, ln), c in per_line.items():
        per_func[func_of(f, ln)].update(c)
    tot = collections.Counter()
    print(L[start].split(':')[0])
    for n, c in sorted(per_func.items(), key=lambda x: -(x[1]['valu'] + x[1]['valu64'] + x[1]['trans'])):
        if sum(c.values()):
            print(f"  {n:28s} valu {c['valu']:5d}  f64 {c['valu64']:4d}  trans {c['trans']:4d}  salu {c['salu']:5d}  lds {c['lds']:4d}  vmem {c['vmem']:4d}")
        tot.update(c)
    print('  total', dict(tot))
    print(f'top {a.top} source lines by vector instructions:')
    for (f, ln), c in sorted(per_line.items(), key=lambda x: -(x[1]['valu'] + x[1]['valu64'] + x[1]['trans']))[:a.top]:
        text = src[f][ln - 1].strip()[:100] if f in src and 0 < ln <= len(src[f]) else ''
        print(f"  {c['valu'] + c['valu64'] + c['trans']:4d}  {f}:{ln:<5d} {text}")
    if a.function:
        print(f'lines of {a.function}:')
        for (f, ln), c in sorted(per_line.items(), key=lambda x: x[0][1]):
            if func_of(f, ln) == a.function:
                print(f'  {ln:5d} {dict(c)}  {src[f][ln - 1].strip()[:100] if f in src else ""}')


if __name__ == '__main__':
    main()
