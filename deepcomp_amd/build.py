"""Builds libdcomp_hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc.

    python -m deepcomp_amd.build [--force] [--jobs N]

One object per base-station count listed in csrc/dcomp_blist.h (all UE-group widths inside), compiled
in parallel, plus the API object; linked into deepcomp_amd/csrc/libdcomp_hip.so.  hipcc cross-compiles
for gfx950 without a GPU present.
"""
import argparse
import concurrent.futures
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, 'build')
LIB = os.path.join(CSRC, 'libdcomp_hip.so')
ARCH = 'gfx950'
# fast-honor-pragmas: `#pragma clang fp contract(off)` in the FP64 movement code must win (bit-exact positions)
# offload-compress: the gfx950 code objects (32 station counts x 7 lane widths x the kernel variants) are stored compressed: 98 -> ~20 MB
CXXFLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast-honor-pragmas', '--offload-compress',
            '-Wall', '-Wno-unused-function']


def b_list():
    if os.environ.get('DCOMP_BUILD_B'):        # development builds: only these base-station counts (e.g. "5,10,32")
        return [int(x) for x in os.environ['DCOMP_BUILD_B'].split(',')]
    txt = open(os.path.join(CSRC, 'dcomp_blist.h')).read()
    line = re.search(r'#define DCOMP_B_LIST\(X\)((?:.*\\\n)*.*)', txt).group(1)
    return [int(x) for x in re.findall(r'X\((\d+)\)', line)]


def _sources():
    return [os.path.join(CSRC, f) for f in ('dcomp_device.h', 'dcomp_wide.h', 'dcomp_dyn.h', 'dcomp_blist.h', 'dcomp_inst.hip', 'dcomp_api.hip', 'dcomp_fragment.h',
                                            'dcomp_big.h', 'dcomp_big.hip')] + \
        [os.path.join(os.path.dirname(HERE), 'include', f) for f in ('dcomp.h', 'dcomp_types.h')]


def _dev_flags():
    if not os.environ.get('DCOMP_BUILD_B'):
        return []
    bl = b_list()
    return ['-DDCOMP_B_LIST(X)=' + ' '.join(f'X({b})' for b in bl), '-DDCOMP_B_LIST_STR="' + ','.join(map(str, bl)) + ' (development build)"']


STAMP = LIB + '.stamp'


def _fingerprint(extra_flags=()):
    """Content hash of every source + the flags: an mtime test calls a library built from OTHER sources (a `git checkout`
    of older files, an experiment that was reverted) up to date."""
    h = hashlib.sha256()
    for f in _sources():
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    h.update(' '.join(CXXFLAGS + list(extra_flags)).encode())
    return h.hexdigest()


KERNEL_SOURCES = ('dcomp_device.h', 'dcomp_wide.h', 'dcomp_dyn.h', 'dcomp_blist.h', 'dcomp_inst.hip')


def kernel_fingerprint(read=None):
    """Hash of what the step / reset / rollout KERNELS are compiled from (the per-station-count objects: csrc/dcomp_inst.hip and
    the headers it includes, + flags) -- not the host side of the ABI (dcomp_api.hip, include/dcomp.h), whose edits cannot change
    a kernel's HBM traffic.  bench.py accepts a tracked --pmc profile while this hash AND the name of the dispatched kernel
    (dcomp_step_kernel_name) match.  read(name) -> bytes: fingerprint of another tree (tools/register_traffic.py --backfill)."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), 'rb').read() if read is None else read(f))
    h.update(' '.join(CXXFLAGS + _dev_flags()).encode())
    return h.hexdigest()


def generic_fingerprint(read=None):
    """The generic kernel (csrc/dcomp_big.h, 33 ... 64 stations / 257 ... 1 024 UEs) on top of kernel_fingerprint(): a `big_kernel` entry of
    profiles/traffic.json is valid only while this matches too (the specialised kernels' entries do not depend on dcomp_big.h)."""
    h = hashlib.sha256(kernel_fingerprint(read).encode())
    for f in ('dcomp_big.h', 'dcomp_big.hip'):
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), 'rb').read() if read is None else read(f))
    return h.hexdigest()


def source_fingerprint():
    """Hash of the kernel sources + flags of the product build: profiles record it (tools/summarize_prof.py), bench.py compares
    it with the library it runs -- a PMC figure taken from other sources is reported as stale.  Needs no git."""
    return _fingerprint(_dev_flags())


def up_to_date(extra_flags=()):
    extra_flags = [f for f in extra_flags if f not in _dev_flags()] + _dev_flags()
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return False
    return open(STAMP).read().strip() == _fingerprint(extra_flags)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError('command failed: ' + ' '.join(cmd) + '\n' + r.stdout)
    return r.stdout


def build(force=False, jobs=None, extra_flags=()):
    extra_flags = [f for f in extra_flags if f not in _dev_flags()] + _dev_flags()
    if not force and up_to_date(extra_flags):
        return LIB
    hipcc = os.environ.get('HIPCC', 'hipcc')
    stamp_at_start = _fingerprint(extra_flags)      # of the sources as they are NOW: an edit during the (long) build must not be stamped as built
    os.makedirs(OBJ, exist_ok=True)
    jobs = jobs or os.cpu_count() or 4
    tasks = []
    for b in b_list():
        o = os.path.join(OBJ, f'dcomp_inst_b{b}.o')
        tasks.append((o, [hipcc] + CXXFLAGS + list(extra_flags) + [f'-DDCOMP_B={b}', '-c', os.path.join(CSRC, 'dcomp_inst.hip'),
                                                                   '-o', o]))
    o_api = os.path.join(OBJ, 'dcomp_api.o')
    tasks.append((o_api, [hipcc] + CXXFLAGS + list(extra_flags) + ['-c', os.path.join(CSRC, 'dcomp_api.hip'), '-o', o_api]))
    o_big = os.path.join(OBJ, 'dcomp_big.o')        # the generic kernel for 33 ... 64 stations: one object for every station count
    tasks.append((o_big, [hipcc] + CXXFLAGS + list(extra_flags) + ['-c', os.path.join(CSRC, 'dcomp_big.hip'), '-o', o_big]))

    def obj_stamp(t):
        """An object is rebuilt when its command line or one of ITS inputs changed: the per-station-count objects do not
        include dcomp_api.hip, dcomp_fragment.h or include/dcomp.h -- the entry-point declarations; they see include/dcomp_types.h
        only -- so an ABI-side edit recompiles one file and relinks, seconds instead of ten minutes."""
        h = hashlib.sha256(' '.join(t[1]).encode())
        for f in _sources():
            if f.endswith(('dcomp_api.hip', 'dcomp_fragment.h', os.sep + 'dcomp.h')) and not t[0].endswith('dcomp_api.o'):
                continue
            if f.endswith(('dcomp_big.h', 'dcomp_big.hip')) and not t[0].endswith(('dcomp_api.o', 'dcomp_big.o')):
                continue
            h.update(os.path.basename(f).encode())
            h.update(open(f, 'rb').read())
        return h.hexdigest()

    def compile_one(t):
        stamp, want = t[0] + '.stamp', obj_stamp(t)
        if not force and os.path.exists(t[0]) and os.path.exists(stamp) and open(stamp).read().strip() == want:
            return ''
        out = _run(t[1])
        with open(stamp, 'w') as f:
            f.write(want + '\n')
        return out
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        for out in ex.map(compile_one, tasks):
            if out.strip():
                sys.stderr.write(out)
    _run([hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + [t[0] for t in tasks] + ['-lpthread'])
    with open(STAMP, 'w') as f:
        f.write(stamp_at_start + '\n')
    return LIB


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--force', action='store_true')
    ap.add_argument('--jobs', type=int, default=None)
    ap.add_argument('--no-dpp', action='store_true', help='build the __shfl_xor fallback reductions (debug)')
    a = ap.parse_args()
    print(build(force=a.force, jobs=a.jobs, extra_flags=['-DDCOMP_NO_DPP'] if a.no_dpp else []))
