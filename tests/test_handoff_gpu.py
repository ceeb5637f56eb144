"""GPU tests of the learner hand-off path: bench.py's RCCL all-gather of rollout fragments, run as real
processes.  The build box has ONE GPU: a single-rank RCCL communicator exercises the whole code path (NCCL init with
device_id, side stream, fragment buffers, all_gather_into_tensor, handoff record); two ranks sharing the device is attempted
and, if RCCL refuses it (duplicate GPU), the reason is recorded instead.  The 1 -> 8 GPU curve is the driver's to measure."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ['--envs', '2048', '--steps', '20', '--warmup', '10', '--no-cpu-baseline', '--no-also', '--no-stream', '--eps-length', '10']


def _free_port():
    """Below the kernel's ephemeral range: a bind(0) port can be handed to any outgoing connection before the launcher binds it."""
    import random
    for _ in range(64):
        p = random.randrange(20000, 32000)
        s = socket.socket()
        try:
            s.bind(('127.0.0.1', p))
            return p
        except OSError:
            continue
        finally:
            s.close()
    raise RuntimeError('no free port')


def _run(cmd):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    r = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    return r.returncode, r.stdout, (json.loads(lines[-1]) if lines else None)


@pytest.mark.parametrize('mode', ['obs', 'summary'])
def test_bench_handoff_over_rccl_single_rank(mode):
    rc, out, j = _run([sys.executable, 'bench.py', '--gpus', '1', '--force-dist', '--gather', mode, '--fragment', '5'] + COMMON)
    assert rc == 0 and j is not None, out[-3000:]
    h = j['handoff']
    assert h['mode'] == mode and h['backend'] == 'rccl' and h['rccl_ranks'] == 1 and h['overlapped_on_side_stream']
    assert j['n_gpus'] == 1 and j['value'] > 0 and j['roofline']['kernel_ms'] <= 1.5 * j['ms_per_step']
    if mode == 'obs':
        assert h['fragments'] == 4 and h['fragment_steps'] == 5
        assert h['bytes_sent_per_rank_per_fragment'] == 5 * 2048 * 32 * (41 + 1) * 4
        assert h['compute_stream_stall_ms_total'] >= 0.0
    else:
        p = j['also']['obs_handoff_probe']             # the obs hand-off is measured next to the summary mode
        assert p['rccl_ranks'] == 1 and p['ms_per_fragment_with_overlapped_all_gather'] > 0


def test_bench_two_ranks_sharing_the_device():
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), 'bench.py', '--gpus', '2', '--same-device', '--gather', 'obs', '--fragment', '5'] + COMMON
    rc, out, j = _run(cmd)
    if rc != 0 or j is None:
        why = [l for l in out.splitlines() if 'uplicate' in l or 'invalid usage' in l or 'NCCL' in l or 'RCCL' in l][:5]
        os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(REPO, 'gpurun_out', 'rccl_two_ranks_one_device.txt'), 'w') as f:
            f.write('two RCCL ranks on one device were refused:\n' + '\n'.join(why) + '\n---- tail\n' + out[-2000:])
        pytest.skip('RCCL refuses two ranks on one device: ' + ' | '.join(why)[:300])
    assert j['handoff']['rccl_ranks'] == 2 and j['n_gpus'] == 2 and j['handoff']['fragments'] == 4


def test_bench_spawns_its_own_ranks_n2_gloo_same_device():
    """`python bench.py --gpus 2` with NO launcher and no WORLD_SIZE in the environment (the form the driver uses for N = 1):
    bench.py starts its two ranks itself.  Both share cuda:0 here, so the process group is gloo (RCCL refuses two ranks on one
    device) -- everything else is the N > 1 control flow: global env ids per rank, RolloutGather fragments, the MAX-reduce of the
    elapsed time, the multi-GPU BASELINE splits in `also`, ONE JSON line from rank 0."""
    env_clean = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--backend', 'gloo', '--same-device', '--gather', 'obs', '--fragment', '5',
           '--envs', '2048', '--steps', '20', '--warmup', '10', '--no-cpu-baseline', '--no-stream', '--eps-length', '10']
    r = subprocess.run(cmd, cwd=REPO, env=env_clean, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert r.returncode == 0 and lines, (r.stdout[-2000:], r.stderr[-3000:])
    j = json.loads(lines[-1])                                   # the JSON line is the LAST line of stdout
    assert j['n_gpus'] == 2 and j['handoff']['rccl_ranks'] == 2 and j['handoff']['backend'] == 'gloo'
    assert j['handoff']['mode'] == 'obs' and j['handoff']['fragments'] == 4
    assert j['handoff']['bytes_received_per_rank_per_fragment'] == 2 * 5 * 2048 * 32 * (41 + 1) * 4
    assert j['value'] > 0 and j['config']['parallelism'].startswith('env-shard x2')
    assert j['handoff']['collectives_in_timed_region'] == 2 * 4        # obs + reward per fragment, 4 fragments in the timed 20 steps
    assert j['handoff']['bytes_in_timed_region']['sent_per_rank'] == 4 * 5 * 2048 * 32 * (41 + 1) * 4
    c4 = j['also']['config4_strong_262144x32x10']                      # strong scaling: the BASELINE totals split over the ranks
    c5 = j['also']['config5_strong_32768x128x32']
    assert c4['n_gpus'] == 2 and c4['total_envs'] == 262144 and c4['envs_per_gpu'] == 131072 and c4['value'] > 0, c4
    assert c5['n_gpus'] == 2 and c5['total_envs'] == 32768 and c5['envs_per_gpu'] == 16384 and c5['value'] > 0, c5


def test_driver_form_n2_has_a_collective_inside_the_timed_region():
    """with the driver's exact `--steps 20 --warmup 5` and an episode of 100 steps NO collective fell inside the timed region,
    while config.collective claimed one per episode.  The default hand-off now happens every min(L, max(4, K)) = 20 steps (phase-shifted by half a period), is
    counted where it is issued, and every rank runs the same pre-warm at every N.  Two self-spawned ranks on one device (gloo: RCCL
    refuses two ranks on one GPU), default workload, secondary configurations on -- the driver's command line otherwise."""
    env_clean = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--steps', '20', '--warmup', '5', '--backend', 'gloo', '--same-device']
    r = subprocess.run(cmd, cwd=REPO, env=env_clean, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert r.returncode == 0 and lines, (r.stdout[-2000:], r.stderr[-3000:])
    j = json.loads(lines[-1])
    h = j['handoff']
    assert j['n_gpus'] == 2 and j['steps'] == 20 and j['warmup'] == 5 and h['mode'] == 'summary' and h['rccl_ranks'] == 2
    assert h['period_steps'] == 20 and h['collectives_in_timed_region'] == 1              # one [E, U + 1] tensor after timed step 10
    assert h['bytes_in_timed_region']['sent_per_rank'] == 4 * 65536 * (32 + 1)
    assert h['bytes_in_timed_region']['received_per_rank'] == 2 * h['bytes_in_timed_region']['sent_per_rank']
    assert '1 collective(s) inside the timed region' in j['config']['collective'] and 'every rank' in j['config']['prewarm']
    a = j['also']
    assert a['measured'].startswith('before the warm-up') and 'config2_4096x10x5_central_fused_rollout' in a     # the pre-warm ran at N = 2 too
    assert a['config4_strong_262144x32x10']['envs_per_gpu'] == 131072 and a['config5_strong_32768x128x32']['envs_per_gpu'] == 16384
    assert set(a['config5_per_gpu_share_of_32768x128x32']) == {'N1_32768_envs', 'N2_16384_envs', 'N4_8192_envs', 'N8_4096_envs'}
    r50 = a['central_65536x10x5']['through_rollout_T50']
    assert r50['frac_of_hbm_peak_kernel_traffic'] < r50['frac_of_hbm_peak_algorithmic']   # both figures, the honest one is the smaller


def test_single_rank_rccl_driver_form_counts_its_collectives():
    """`--gpus 1 --spawn --steps 20 --warmup 5`: a real RCCL communicator (one rank); the summary hand-off runs inside the timed region."""
    env_clean = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '1', '--spawn', '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-also'],
                       cwd=REPO, env=env_clean, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    j = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    h = j['handoff']
    assert h['backend'] == 'rccl' and h['rccl_ranks'] == 1 and h['collectives_in_timed_region'] == 1 and h['period_steps'] == 20


def test_bench_spawns_its_own_ranks_summary_mode_probe():
    env_clean = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--backend', 'gloo', '--same-device', '--fragment', '5', '--no-also'] + COMMON
    r = subprocess.run(cmd, cwd=REPO, env=env_clean, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    j = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert j['handoff']['mode'] == 'summary' and j['handoff']['rccl_ranks'] == 2
    p = j['also']['obs_handoff_probe']                          # the observation hand-off next to the summary mode
    assert p['rccl_ranks'] == 2 and p['exposed_handoff_ms_per_fragment'] >= 0 and p['bytes_received_per_rank_per_fragment'] == 2 * p['bytes_sent_per_rank_per_fragment']


def test_bench_self_spawn_single_rank_rccl():
    """The self-spawning form with one rank: a real RCCL communicator of size 1 (`--spawn` forces the spawn path that
    `--gpus N > 1` takes by itself)."""
    env_clean = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '1', '--spawn', '--gather', 'summary'] + COMMON, cwd=REPO, env=env_clean,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    j = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert j['n_gpus'] == 1 and j['handoff']['rccl_ranks'] == 1 and j['handoff']['backend'] == 'rccl'


@pytest.mark.parametrize('mode', ['obs', 'summary'])
def test_bench_direct_all_gather_spelled_out_two_gloo_ranks(mode):
    """`--gather-algo p2p`: every rank posts one send to, and one receive from, each peer in one batch (RolloutGather(algo='p2p'),
    SURVEY.md 8e "direct (one-shot, all-peers) all-gather rather than ring") -- same fragments, same byte counts, same handle as the
    collective form; the record names the algorithm."""
    rc, out, j = _run([sys.executable, 'bench.py', '--gpus', '2', '--backend', 'gloo', '--same-device', '--gather', mode, '--gather-algo', 'p2p',
                       '--fragment', '5'] + COMMON)
    assert rc == 0 and j is not None, out[-3000:]
    h = j['handoff']
    assert h['algo'] == 'p2p' and h['rccl_ranks'] == 2 and h['mode'] == mode and j['value'] > 0 and h['collectives_in_timed_region'] > 0
    assert h['bytes_in_timed_region']['received_per_rank'] == 2 * h['bytes_in_timed_region']['sent_per_rank'] > 0
    if mode == 'obs':
        assert h['fragments'] == 4 and h['bytes_sent_per_rank_per_fragment'] == 5 * 2048 * 32 * (41 + 1) * 4
        assert h['bytes_received_per_rank_per_fragment'] == 2 * h['bytes_sent_per_rank_per_fragment']


def test_bench_rccl_direct_hints_are_set_before_the_communicator_exists():
    """`--rccl-direct`: the hints of sharded.rccl_direct_hints() are in the rank's environment when init_process_group runs, and the
    record carries what was set (a real RCCL communicator of size 1: the box has one GPU)."""
    env_clean = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                                                     'RCCL_DIRECT_ALLGATHER_THRESHOLD', 'NCCL_PROTO')}
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '1', '--spawn', '--rccl-direct', '--gather', 'obs', '--fragment', '5'] + COMMON, cwd=REPO,
                       env=env_clean, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    j = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert j['handoff']['backend'] == 'rccl' and j['handoff']['algo'] == 'collective'
    assert j['handoff']['rccl_hints'] == {'RCCL_DIRECT_ALLGATHER_THRESHOLD': str(1 << 34), 'NCCL_PROTO': 'Simple'}


@pytest.mark.parametrize('via_pack', [False, True])
def test_bench_compact_observation_handoff_single_rank_rccl(via_pack):
    """`--gather obs --compact`: the fragment crosses the collective as the lossless compact record, U (B + 2) + 2B words per env-step
    instead of U (4B + 1) -- written by the steps themselves (dcomp_out.obs_compact) or, `--compact-via-pack`, by dcomp_pack_fragment
    inside the timed region."""
    rc, out, j = _run([sys.executable, 'bench.py', '--gpus', '1', '--force-dist', '--gather', 'obs', '--compact', '--fragment', '5'] +
                      (['--compact-via-pack'] if via_pack else []) + COMMON)
    assert rc == 0 and j is not None, out[-3000:]
    h = j['handoff']
    assert h['mode'] == 'obs' and h['compact'] is True and h['fragments'] == 4
    assert ('dcomp_pack_fragment' in h['compact_written_by']) == via_pack
    words = 32 * (10 + 2) + 2 * 10
    assert h['bytes_sent_per_rank_per_fragment'] == 5 * 2048 * (words + 32) * 4
    assert h['bytes_in_timed_region']['sent_per_rank'] == 4 * h['bytes_sent_per_rank_per_fragment']
    raw = 5 * 2048 * 32 * (41 + 1) * 4
    assert raw / h['bytes_sent_per_rank_per_fragment'] > 3.0


def test_bench_compact_handoff_two_gloo_ranks_same_device():
    """Two self-spawned ranks (gloo, one device): every fragment is packed on the GPU, staged through the host as int32 words, gathered
    and counted -- the N > 1 control flow of `--gather obs --compact`."""
    env_clean = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--backend', 'gloo', '--same-device', '--gather', 'obs', '--compact', '--fragment', '5',
           '--no-also'] + COMMON
    r = subprocess.run(cmd, cwd=REPO, env=env_clean, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    j = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    h = j['handoff']
    words = 32 * 12 + 20
    assert h['compact'] is True and h['rccl_ranks'] == 2 and h['fragments'] == 4 and h['collectives_in_timed_region'] == 8
    assert h['bytes_sent_per_rank_per_fragment'] == 5 * 2048 * (words + 32) * 4
    assert h['bytes_received_per_rank_per_fragment'] == 2 * h['bytes_sent_per_rank_per_fragment']


def test_bench_under_torch_distributed_run_two_gloo_ranks():
    """The driver's launcher form for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W`), with the process group on gloo because both ranks share this box's one
    GPU: RANK / LOCAL_RANK / WORLD_SIZE come from the launcher, rank 0 prints the one JSON line."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), 'bench.py', '--gpus', '2', '--steps', '20', '--warmup', '5', '--backend', 'gloo', '--same-device',
           '--envs', '4096', '--no-also', '--no-stream']
    rc, out, j = _run(cmd)
    assert rc == 0 and j is not None, out[-3000:]
    assert j['n_gpus'] == 2 and j['steps'] == 20 and j['config']['parallelism'].startswith('env-shard x2') and j['config']['envs_per_gpu'] == 4096
    assert j['handoff']['rccl_ranks'] == 2 and j['handoff']['collectives_in_timed_region'] == 1
    assert len([l for l in out.splitlines() if l.startswith('{"metric"')]) == 1


def test_driver_form_n8_gloo_same_device():
    """The rank count of the node the scaling curve will be drawn on: `--gpus 8 --steps 20 --warmup 5`, eight
    self-spawned ranks sharing this box's one GPU over gloo, a small batch per rank, secondary configurations ON -- so the two
    multi-GPU BASELINE configurations run as what they are at N = 8: 32 768 x 32 x 10 and 4 096 x 128 x 32 per rank.  ONE JSON line;
    the hand-off counted inside the timed region; every rank's own elapsed time and CPU placement in the line; the
    rollout hand-off throughput next to the summary's at top level; and the whole run well inside the driver's patience."""
    import time
    env_clean = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, 'bench.py', '--gpus', '8', '--steps', '20', '--warmup', '5', '--backend', 'gloo', '--same-device', '--envs', '256']    # (small: 24 gloo all-gathers of 8 fragments each go through loopback TCP)
    t0 = time.time()
    r = subprocess.run(cmd, cwd=REPO, env=env_clean, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    wall = time.time() - t0
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert r.returncode == 0 and lines, (r.stdout[-2000:], r.stderr[-3000:])
    assert len([l for l in lines if l.startswith('{"metric"')]) == 1 and lines[-1].startswith('{"metric"')
    j = json.loads(lines[-1])
    print(f'8 gloo ranks on one device: wall {wall:.0f} s, rank 0 wall_s {j["wall_s"]:.0f} s')
    assert wall < 600 and j['wall_s'] < 600
    h = j['handoff']
    assert j['n_gpus'] == 8 and j['steps'] == 20 and j['warmup'] == 5 and j['scaling'] == 'weak' and j['config']['envs_per_gpu'] == 256
    assert h['rccl_ranks'] == 8 and h['mode'] == 'summary' and h['collectives_in_timed_region'] >= 1
    assert h['bytes_in_timed_region']['received_per_rank'] == 8 * h['bytes_in_timed_region']['sent_per_rank'] > 0
    # every rank's own clock, and where each rank ran
    e = j['elapsed_per_rank_ms']
    assert len(e['ranks']) == 8 and e['min'] <= e['max'] and e['max'] == pytest.approx(j['ms_per_step'] * 20, rel=1e-6)
    pl = j['config']['rank_placement']
    assert len(pl) == 8 and all('pinned' in x and ('cpus' in x or 'why_not' in x) for x in pl)
    # the rollout hand-off (observations to every rank) next to the summary hand-off `value` carries
    w = h['with_rollout_handoff']
    for k in ('rows', 'compact_record_packed_after_the_steps', 'compact_record_written_by_the_steps'):
        assert w[k]['env_steps_per_s'] > 0 and w[k]['bytes_received_per_rank_per_fragment'] > 0, k
    assert w['rows']['bytes_received_per_rank_per_fragment'] == 8 * 4 * 256 * 32 * (41 + 1) * 4        # 4-step fragments from 8 ranks
    assert w['compact_record_written_by_the_steps']['bytes_received_per_rank_per_fragment'] < w['rows']['bytes_received_per_rank_per_fragment'] / 3
    # the N = 8 points of the two strong-scaling curves ARE the BASELINE configurations 4 and 5
    c4, c5 = j['also']['config4_strong_262144x32x10'], j['also']['config5_strong_32768x128x32']
    assert c4['n_gpus'] == 8 and c4['envs_per_gpu'] == 32768 and c4['total_envs'] == 262144 and c4['value'] > 0, c4
    assert c5['n_gpus'] == 8 and c5['envs_per_gpu'] == 4096 and c5['total_envs'] == 32768 and c5['value'] > 0, c5
