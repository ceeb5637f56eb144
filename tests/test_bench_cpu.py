"""CPU tests of bench.py's host-side pieces: NUMA placement of a rank against a planted sysfs tree, the
CPU-list helpers, the CPU baseline record (single core next to all cores)."""
import os
import sys
import types

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def _fake_torch(domain, bus, device):
    props = types.SimpleNamespace(pci_domain_id=domain, pci_bus_id=bus, pci_device_id=device)
    return types.SimpleNamespace(cuda=types.SimpleNamespace(get_device_properties=lambda i: props))


def _plant(tmp_path, bdf, node, cpulist):
    d = tmp_path / 'bus' / 'pci' / 'devices' / bdf
    d.mkdir(parents=True)
    (d / 'numa_node').write_text(f'{node}\n')
    if node >= 0:
        n = tmp_path / 'devices' / 'system' / 'node' / f'node{node}'
        n.mkdir(parents=True)
        (n / 'cpulist').write_text(cpulist + '\n')
    return str(tmp_path)


def test_cpulist_round_trip():
    assert bench._parse_cpulist('0-3,8,10-11\n') == {0, 1, 2, 3, 8, 10, 11}
    assert bench._fmt_cpulist({0, 1, 2, 3, 8, 10, 11}) == '0-3,8,10-11'
    assert bench._parse_cpulist('') == set() and bench._fmt_cpulist(set()) == ''


def test_rank_is_bound_to_the_cpus_of_its_gpus_numa_node(tmp_path):
    before = os.sched_getaffinity(0)
    if len(before) < 2:
        pytest.skip('one CPU only')
    keep = sorted(before)[:max(1, len(before) // 2)]
    sysfs = _plant(tmp_path, '0000:c5:00.0', 1, bench._fmt_cpulist(set(keep)) + ',4090-4095')     # CPUs outside the mask are dropped
    try:
        info = bench.pin_to_gpu_numa_node(_fake_torch(0, 0xc5, 0), 3, sysfs=sysfs)
        assert info['pinned'] and info['numa_node'] == 1 and info['pci'] == '0000:c5:00.0' and info['local_rank'] == 3
        assert os.sched_getaffinity(0) == set(keep) and info['cpus'] == bench._fmt_cpulist(set(keep))
    finally:
        os.sched_setaffinity(0, before)


def test_no_numa_node_or_no_sysfs_leaves_the_rank_alone(tmp_path):
    before = os.sched_getaffinity(0)
    info = bench.pin_to_gpu_numa_node(_fake_torch(0, 5, 0), 0, sysfs=_plant(tmp_path / 'a', '0000:05:00.0', -1, ''))
    assert not info['pinned'] and info['numa_node'] == -1 and 'no NUMA node' in info['why_not']
    info = bench.pin_to_gpu_numa_node(_fake_torch(0, 6, 0), 0, sysfs=str(tmp_path / 'missing'))
    assert not info['pinned'] and 'why_not' in info                       # an error is recorded, never raised
    sysfs = _plant(tmp_path / 'b', '0000:07:00.0', 0, bench._fmt_cpulist(before))
    info = bench.pin_to_gpu_numa_node(_fake_torch(0, 7, 0), 0, enable=False, sysfs=sysfs)
    assert not info['pinned'] and info['why_not'] == '--no-pin' and info['cpus']
    assert os.sched_getaffinity(0) == before


def test_cpu_baseline_reports_single_core_and_the_link_to_the_reference():
    from deepcomp_amd import scenarios
    scn = scenarios.grid_map(5, 'mixed').with_ues(num_slow=10)
    r = bench.cpu_baseline(scn, 'central', 10, 5, budget_s=0.5)
    assert r['kind'] == 'port' and r['value'] > 0 and r['cores'] >= 1
    s = r['single_core']
    assert s['cores'] == 1 and s['value'] > 0 and s['all_cores_over_single_core'] == pytest.approx(r['value'] / s['value'])
    ref = r['reference_step']
    assert ref['kind'] == 'reference' and ref['value'] == 398.0           # CentralRelNormEnv 10 x 5 mixed, 1 core
    assert ref['port_single_core_over_reference_step'] == pytest.approx(s['value'] / 398.0)


def test_tracked_traffic_goes_stale_with_the_sources_it_was_profiled_on(monkeypatch):
    """roofline.traffic is a tracked PMC figure, valid only for the kernel sources it was taken from: the specialised kernels' entries
    hang on build.kernel_fingerprint, a `big_kernel` entry also on csrc/dcomp_big.h (build.generic_fingerprint)."""
    from deepcomp_amd import build
    for key, kern in (('65536x32x10_multi_mixed', 'step_kernel<10, 32, 2>'), ('8192x32x64_multi_mixed', 'big_kernel<32, false>')):
        nbytes, src = bench.traffic_from_profile(key, kern)
        assert nbytes and nbytes > 1e8, (key, src)                     # the committed entries belong to the committed sources
        assert bench.traffic_from_profile(key, kern + ' ')[0] is None   # another instantiation is dispatched
    monkeypatch.setattr(build, 'generic_fingerprint', lambda read=None: 'edited')
    monkeypatch.setattr(build, 'source_fingerprint', lambda: 'edited')
    assert bench.traffic_from_profile('65536x32x10_multi_mixed', 'step_kernel<10, 32, 2>')[0]        # untouched by an edit of dcomp_big.h
    nbytes, src = bench.traffic_from_profile('8192x32x64_multi_mixed', 'big_kernel<32, false>')
    assert nbytes is None and 'STALE' in src and 'dcomp_big.h' in src
    monkeypatch.setattr(build, 'kernel_fingerprint', lambda read=None: 'edited')
    assert bench.traffic_from_profile('65536x32x10_multi_mixed', 'step_kernel<10, 32, 2>')[0] is None
