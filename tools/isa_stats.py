#!/usr/bin/env python3
"""Static ISA statistics of the gfx950 kernels of one translation unit (no GPU needed): registers, spills, and -- per kernel --
instruction counts by class, for the whole kernel and for its outermost loop (the step loop of the fused rollout).

    python tools/isa_stats.py --b 5 --upad 16 [--filter rollout] [--flags=-DDCOMP_X=1] [--keep /tmp/isa]

Compiles deepcomp_amd/csrc/dcomp_inst.hip to assembly (hipcc -S --cuda-device-only, a few seconds with --upad).  Static counts are
not dynamic counts (rare branches are in there), but a change that removes spills or a block of VALU work shows up here
before any GPU minute is spent."""
import argparse
import collections
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(op):
    if op.startswith('v_readlane') or op.startswith('v_writelane'):
        return 'lane'
    if op.startswith(('v_log', 'v_exp', 'v_rcp', 'v_rsq', 'v_sqrt')):
        return 'trans'
    if op.startswith('v_') and ('f64' in op or op.startswith('v_mov_b64')):
        return 'valu64'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith('s_nop'):
        return 'nop'
    if op.startswith(('s_cbranch', 's_branch')):
        return 'branch'
    if op.startswith('s_load') or op.startswith('s_buffer_load'):
        return 'sload'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    return 'other'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--b', type=int, default=5)
    ap.add_argument('--upad', type=int, default=16)
    ap.add_argument('--filter', default='')
    ap.add_argument('--flags', default='')
    ap.add_argument('--keep', default='/tmp/isa')
    a = ap.parse_args()
    os.makedirs(a.keep, exist_ok=True)
    out = os.path.join(a.keep, f'b{a.b}_u{a.upad}.s')
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast-honor-pragmas', f'-DDCOMP_B={a.b}',
           '-S', '--cuda-device-only', '-o', out, os.path.join(REPO, 'deepcomp_amd', 'csrc', 'dcomp_inst.hip')]
    if a.upad:
        cmd.insert(-4, f'-DDCOMP_ONLY_UPAD={a.upad}')
    cmd[1:1] = [f for f in a.flags.split() if f]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        sys.exit(r.stdout)
    txt = open(out).read()
    meta = {}
    for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', txt):
        meta[m.group(1)] = tuple(int(x) for x in m.groups()[1:])
    lds = {m.group(1): int(m.group(2)) for m in re.finditer(r'\.group_segment_fixed_size:\s+(\d+)', '')}
    for name in sorted(meta):
        dem = subprocess.run(['c++filt', name], stdout=subprocess.PIPE, text=True).stdout.strip()
        if a.filter and a.filter not in dem:
            continue
        body = txt[txt.index('\n' + name + ':'):]
        body = body[:body.index('s_endpgm') + 8].splitlines()
        loop_at = next((i for i, l in enumerate(body) if 'Loop Header: Depth=1' in l), None)
        def count(lines):
            c = collections.Counter()
            for l in lines:
                l = l.strip()
                if not l or l.startswith((';', '.', '_Z')) or l.endswith(':'):
                    continue
                c[classify(l.split()[0])] += 1
            return c
        whole = count(body)
        s, sp, v, vsp = meta[name]
        print(f'{dem}\n   sgpr {s} (spilled {sp})  vgpr {v} (spilled {vsp})')
        print('   whole kernel:', dict(sorted(whole.items())), 'total', sum(whole.values()))
        if loop_at is not None:
            lp = count(body[loop_at:])
            print('   from the first loop header on:', dict(sorted(lp.items())), 'total', sum(lp.values()))


if __name__ == '__main__':
    main()
