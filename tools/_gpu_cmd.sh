python bench.py --gpus 1 --spawn --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_line_spawn.json; python - <<'PY'
import json
s=json.load(open('gpurun_out/r04_bench_line_spawn.json')); p=s['also']['obs_handoff_probe']
print(s['value'], s['ms_per_step'], s['handoff']['collectives_in_timed_region'], p['ms_per_fragment_with_overlapped_all_gather'], p['compact_record']['ms_per_fragment_with_overlapped_all_gather'], p['compact_record_from_the_step']['ms_per_fragment_with_overlapped_all_gather'])
PY
