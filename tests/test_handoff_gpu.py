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
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


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
