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
    for key, kern in (('65536x32x10_multi_mixed', 'step_kernel<10, 32, 2>'), ('8192x32x64_multi_mixed', 'big_kernel<32, false, false, false, false, false>')):
        nbytes, src = bench.traffic_from_profile(key, kern)
        assert nbytes and nbytes > 1e8, (key, src)                     # the committed entries belong to the committed sources
        assert bench.traffic_from_profile(key, kern + ' ')[0] is None   # another instantiation is dispatched
    monkeypatch.setattr(build, 'generic_fingerprint', lambda read=None: 'edited')
    monkeypatch.setattr(build, 'source_fingerprint', lambda: 'edited')
    assert bench.traffic_from_profile('65536x32x10_multi_mixed', 'step_kernel<10, 32, 2>')[0]        # untouched by an edit of dcomp_big.h
    nbytes, src = bench.traffic_from_profile('8192x32x64_multi_mixed', 'big_kernel<32, false, false, false, false, false>')
    assert nbytes is None and 'STALE' in src and 'dcomp_big.h' in src
    monkeypatch.setattr(build, 'kernel_fingerprint', lambda read=None: 'edited')
    assert bench.traffic_from_profile('65536x32x10_multi_mixed', 'step_kernel<10, 32, 2>')[0] is None


def test_predict_prints_the_1_to_8_gpu_curve_without_a_gpu():
    """`python bench.py --predict`: the predicted weak- / strong-scaling curve and the throughput with the rollout hand-off,
    from the measured single-GPU figures and the stated link assumptions -- a markdown table and one JSON line, no torch, no GPU.  The table in
    DESIGN.md section 7 is this output."""
    import json
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--predict'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.strip().splitlines()
    pr = json.loads(lines[-1])['predicted']
    rows = {r['n_gpus']: r for r in pr['rows']}
    assert sorted(rows) == [1, 2, 4, 8] and pr['steps'] == 20
    assert rows[1]['weak_speedup'] == 1.0 and 6.0 <= rows[8]['weak_speedup'] <= 8.0          # north_star: >= 6 x at 8 GPUs
    assert all(rows[a]['weak_ms_per_step'] <= rows[b]['weak_ms_per_step'] for a, b in ((1, 2), (2, 4), (4, 8)))
    assert rows[8]['config5_strong_speedup'] > 8.0 > rows[8]['config4_strong_speedup'] > 6.0  # config 5's share fits the Infinity Cache
    # the full-observation hand-off is link-bound at every N > 1: far below stepping, and the compact record moves ~3.1 x fewer bytes
    assert rows[8]['with_rollout_handoff_rows_env_steps_per_s'] < 0.05 * rows[8]['weak_value_env_steps_per_s']
    assert 2.9 < rows[8]['with_rollout_handoff_compact_env_steps_per_s'] / rows[8]['with_rollout_handoff_rows_env_steps_per_s'] < 3.3
    table = [l for l in lines if l.startswith('|')]
    assert len(table) == 6 and table[2].startswith('| 1 |') and table[5].startswith('| 8 |')
    design = open(os.path.join(REPO, 'DESIGN.md')).read()
    for l in table[2:]:
        assert l in design, 'DESIGN.md section 7 does not hold the table `bench.py --predict` prints'
    assert bench.predict(steps=1000)['rows'][3]['weak_speedup'] > rows[8]['weak_speedup']     # a longer timed region dilutes the bracket
