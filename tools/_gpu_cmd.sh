mkdir -p gpurun_out
bash tools/profile_all.sh r04 2>&1 | tail -20
python bench.py --gpus 1 --spawn --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('spawn', j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['handoff']['collectives_in_timed_region'], j['handoff']['host_blocked_ms_total'])"
