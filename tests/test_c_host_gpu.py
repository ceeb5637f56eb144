"""The C ABI from a host that is not Python: examples/c_host_step.cpp is compiled with hipcc against
include/dcomp.h, linked with libdcomp_hip.so and run as its own process -- dcomp_create (the header's guarded macro), caller-allocated
device buffers, dcomp_reset, 12 x dcomp_step on its own stream, dcomp_check -- and the checksums it prints are held byte for byte to the
same 12 steps through deepcomp_amd.env.BatchedMobileEnv (the ctypes binding).  Same library, two independent bindings."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fnv1a(b):
    h = 1469598103934665603
    for x in np.frombuffer(b, dtype=np.uint8).tolist():
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.fixture(scope='module')
def c_host(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp('c_host') / 'c_host_step')
    csrc = os.path.join(REPO, 'deepcomp_amd', 'csrc')
    r = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O2', '-I', os.path.join(REPO, 'include'), '-o', exe, os.path.join(REPO, 'examples', 'c_host_step.cpp'),
                        '-L', csrc, '-ldcomp_hip', f'-Wl,-rpath,{csrc}'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    return exe


@pytest.mark.parametrize('kind', ['central', 'multi'])
def test_c_host_and_python_binding_produce_the_same_bytes(c_host, kind):
    import torch
    from deepcomp_amd import scenarios
    from deepcomp_amd.entities import build_from_scenario
    from deepcomp_amd.env import BatchedMobileEnv
    r = subprocess.run([c_host, kind], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    got = dict(l.split()[:2] for l in r.stdout.splitlines() if l.split() and l.split()[0] in ('obs', 'reward', 'ue_dr', 'pos', 'conn'))
    assert 'library ABI 3' in r.stdout and 'stale caller: -7 (DCOMP_EABI)' in r.stdout
    E, U, B, T = 64, 6, 4, 12
    m, bs, ues = build_from_scenario(scenarios.custom_map('mixed').with_ues(num_static=1, num_slow=3, num_fast=2))
    env = BatchedMobileEnv(m, bs, ues, kind, num_envs=E, seed=42, rng='philox')
    env.reset()
    lcg = 12345
    for t in range(T):
        a = np.empty(E * U, dtype=np.uint8)
        for i in range(E * U):
            lcg = (lcg * 1664525 + 1013904223) & 0xFFFFFFFF
            a[i] = (lcg >> 24) % (B + 1)
        env.step(torch.from_numpy(a.reshape(E, U)).cuda())
    env.check()
    want = {'obs': env.obs, 'reward': env.reward, 'ue_dr': env.ue_dr, 'pos': env.pos, 'conn': env.conn}
    for k, t in want.items():
        assert int(got[k], 16) == _fnv1a(t.cpu().numpy().tobytes()), f'{k}: the C host and the Python binding disagree'
