#!/usr/bin/env python3
"""Register the HBM traffic of a tracked rocprofv3 --pmc summary in profiles/traffic.json (what bench.py's roofline.traffic reads).

    python tools/register_traffic.py <tag> <workload-key> <kernel-substring>
    python tools/register_traffic.py --backfill <commit>      (kernel_fingerprint for entries profiled at <commit>)
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


def backfill(commit):
    """Entries registered before kernel_fingerprint existed: `git show <commit>:<file>` rebuilds the tree the profile names, its
    FULL fingerprint must equal the entry's source_fingerprint (proof that this is the profiled tree), then the kernel-only
    fingerprint of that same tree is recorded."""
    sys.path.insert(0, REPO)
    from deepcomp_amd import build as b

    def show(path):
        return subprocess.check_output(['git', '-C', REPO, 'show', f'{commit}:{path}'])
    h = b.hashlib.sha256()
    for f in b._sources():
        h.update(os.path.basename(f).encode())
        h.update(show(os.path.relpath(f, REPO)))
    h.update(' '.join(b.CXXFLAGS).encode())
    full, kern = h.hexdigest(), b.kernel_fingerprint(read=lambda name: show('deepcomp_amd/csrc/' + name))
    path = os.path.join(REPO, 'profiles', 'traffic.json')
    db = json.load(open(path))
    n = 0
    for key, ent in db.items():
        if ent.get('source_fingerprint') == full and not ent.get('kernel_fingerprint'):
            ent['kernel_fingerprint'] = kern
            n += 1
    with open(path, 'w') as f:
        json.dump(db, f, indent=1, sort_keys=True)
        f.write('\n')
    print(f'{commit}: full fingerprint {full[:16]}..., kernel fingerprint {kern[:16]}...; {n} entries completed')


def main():
    if sys.argv[1] == '--backfill':
        return backfill(sys.argv[2])
    tag, key, kern = sys.argv[1:4]
    txt = open(os.path.join(REPO, 'profiles', f'{tag}_summary.txt')).read()
    fp = re.search(r'^source_fingerprint: (\S+)', txt, re.M)
    kfp = re.search(r'^kernel_fingerprint: (\S+)', txt, re.M)
    gfp = re.search(r'^generic_fingerprint: (\S+)', txt, re.M)
    vals = {}
    for line in txt.splitlines():
        line = line.strip()
        # (tools/summarize_prof.py cuts the names of the counter rows: a cut name is a prefix of the kernel's)
        if line.startswith('void ') and '{' in line and (kern in line or (len(line[5:line.index('{')].strip()) >= 40 and ('dcomp::' + kern).startswith(line[5:line.index('{')].strip()))):
            d = ast.literal_eval(line[line.index('{'):line.index('}') + 1])
            vals.update(d)
    avg_ns = calls = None
    for line in txt.splitlines():             # "== kernel stats ==" rows: {'Name': 'void dcomp::step_kernel<10, 32, 2>(...)', 'Calls': ..., 'AverageNs': ...}
        line = line.strip()
        if line.startswith("{'Name'") and kern in line:
            d = ast.literal_eval(line)
            avg_ns, calls = float(d['AverageNs']), int(d['Calls'])
            break
    if 'FETCH_SIZE' not in vals or 'WRITE_SIZE' not in vals:
        sys.exit(f'no FETCH_SIZE / WRITE_SIZE rows for {kern!r} in profiles/{tag}_summary.txt')
    try:
        commit = subprocess.check_output(['git', '-C', REPO, 'rev-parse', '--short', 'HEAD'], text=True).strip()
    except Exception:      # noqa: BLE001
        commit = '?'
    path = os.path.join(REPO, 'profiles', 'traffic.json')
    db = json.load(open(path)) if os.path.exists(path) else {}
    old = db.get(key, {})
    if old.get('tag') == tag and old.get('source_fingerprint') == (fp.group(1) if fp else None):
        commit = old.get('commit', commit)         # the same profile registered again (new fields): it still belongs to the tree it was taken from
    db[key] = {'tag': tag, 'kernel': kern, 'fetch_kib': vals['FETCH_SIZE'], 'write_kib': vals['WRITE_SIZE'],
               'source_fingerprint': fp.group(1) if fp else None, 'kernel_fingerprint': kfp.group(1) if kfp else None, 'commit': commit,
               'bytes_per_launch': (2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0,
               'valu_insts_per_launch': vals.get('SQ_INSTS_VALU'), 'waves_per_launch': vals.get('SQ_WAVES'),
               'avg_ns': avg_ns, 'calls': calls,             # rocprofv3 --kernel-trace --stats mean of the same kernel (bench.py roofline.frac_profile)
               'note': 'bytes = (2 x FETCH_SIZE [gfx950 wide-read correction] + WRITE_SIZE) KiB per launch'}
    if kern.startswith('big_kernel'):
        # the generic kernel's own sources (build.generic_fingerprint); a summary from before that line existed belongs to the tree whose FULL
        # fingerprint it names -- if that is this tree, this tree's generic fingerprint is the profile's
        sys.path.insert(0, REPO)
        from deepcomp_amd import build as b
        db[key]['generic_fingerprint'] = gfp.group(1) if gfp else (b.generic_fingerprint() if fp and fp.group(1) == b.source_fingerprint() else None)
    if old.get('tag') == tag and not db[key]['kernel_fingerprint']:
        db[key]['kernel_fingerprint'] = old.get('kernel_fingerprint')
    with open(path, 'w') as f:
        json.dump(db, f, indent=1, sort_keys=True)
        f.write('\n')
    print(key, db[key])


if __name__ == '__main__':
    main()
