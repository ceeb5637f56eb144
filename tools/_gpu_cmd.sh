mkdir -p gpurun_out
python -m pytest tests/test_handoff_gpu.py -m gpu -q 2>&1 | tail -3
python bench.py --gpus 1 --spawn --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; h=j.get('handoff') or {}
print('spawn', round(j['ms_per_step'],4), round(r['kernel_ms'],4), 'gaps', round(r['between_runs_ms_total'],4), h.get('collectives_in_timed_region'), h.get('host_blocked_ms_total'))"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('plain', round(j['ms_per_step'],4), round(r['kernel_ms'],4))"
