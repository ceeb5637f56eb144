#!/usr/bin/env python3
"""Register the HBM traffic of a tracked rocprofv3 --pmc summary in profiles/traffic.json (what bench.py's roofline.traffic reads).

    python tools/register_traffic.py <tag> <workload-key> <kernel-substring>
    e.g.  python tools/register_traffic.py r03a_c3 65536x32x10_multi_mixed 'step_kernel<10, 32, 2>'

Reads profiles/<tag>_summary.txt (written by tools/summarize_prof.py on the GPU box): the per-dispatch FETCH_SIZE / WRITE_SIZE
averages (KiB) of the named kernel and the `source_fingerprint:` line (content hash of the kernel sources the profiled library
was built from).  bench.py reports the traffic only while the library it runs has the same fingerprint."""
import ast
import json
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, key, kern = sys.argv[1:4]
    txt = open(os.path.join(REPO, 'profiles', f'{tag}_summary.txt')).read()
    fp = re.search(r'^source_fingerprint: (\S+)', txt, re.M)
    vals = {}
    for line in txt.splitlines():
        line = line.strip()
        if line.startswith('void ') and kern in line and ("{'FETCH_SIZE'" in line or "{'WRITE_SIZE'" in line):
            d = ast.literal_eval(line[line.index('{'):line.index('}') + 1])
            vals.update(d)
    if 'FETCH_SIZE' not in vals or 'WRITE_SIZE' not in vals:
        sys.exit(f'no FETCH_SIZE / WRITE_SIZE rows for {kern!r} in profiles/{tag}_summary.txt')
    try:
        commit = subprocess.check_output(['git', '-C', REPO, 'rev-parse', '--short', 'HEAD'], text=True).strip()
    except Exception:      # noqa: BLE001
        commit = '?'
    path = os.path.join(REPO, 'profiles', 'traffic.json')
    db = json.load(open(path)) if os.path.exists(path) else {}
    db[key] = {'tag': tag, 'kernel': kern, 'fetch_kib': vals['FETCH_SIZE'], 'write_kib': vals['WRITE_SIZE'],
               'source_fingerprint': fp.group(1) if fp else None, 'commit': commit,
               'bytes_per_launch': (2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0,
               'note': 'bytes = (2 x FETCH_SIZE [gfx950 wide-read correction] + WRITE_SIZE) KiB per launch'}
    with open(path, 'w') as f:
        json.dump(db, f, indent=1, sort_keys=True)
        f.write('\n')
    print(key, db[key])


if __name__ == '__main__':
    main()
