#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel-trace stats + PMC passes) into a small text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:                                    # which kernel sources these numbers belong to (bench.py: roofline.traffic_source)
    from deepcomp_amd import build as _b
    print('source_fingerprint:', _b.source_fingerprint())
    print('kernel_fingerprint:', _b.kernel_fingerprint())
    print('generic_fingerprint:', _b.generic_fingerprint())
except Exception as ex:                 # noqa: BLE001
    print('source_fingerprint: unknown', ex)


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print('== kernel stats (rocprofv3 --kernel-trace --stats) ==')
for f in find('trace/**/*kernel_stats.csv'):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print({k: row[k] for k in row if k in ('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs')})
for f in find('trace_bench.log'):
    print('bench line:', [l for l in open(f) if l.startswith('{')][-1:] )

print('== PMC passes (per-dispatch averages by kernel) ==')
for f in find('pmc_*/**/*counter_collection.csv'):
    agg = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            agg[row['Kernel_Name'][:96]][row['Counter_Name']].append(float(row['Counter_Value']))
    print(os.path.relpath(f, out))
    for k, cs in agg.items():
        if 'step_kernel' not in k and 'reset_kernel' not in k and 'rollout_kernel' not in k and 'big_kernel' not in k:
            continue
        print('  ', k, {c: (sum(v) / len(v)) for c, v in cs.items()}, 'dispatches', len(next(iter(cs.values()))))
