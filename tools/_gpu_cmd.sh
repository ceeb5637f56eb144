python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err; tail -c 300 gpurun_out/r04_bench_line.json; echo
python bench.py --gpus 1 --steps 20 --warmup 5 --compact-step --no-also --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_line_compact_step.json; head -c 1500 gpurun_out/r04_bench_line_compact_step.json; echo
python bench.py --gpus 1 --spawn --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_line_spawn.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_bench_line_spawn.json'))
print(d['value'], d['ms_per_step'], d.get('handoff',{}).get('collectives_in_timed_region'))
p=d.get('handoff',{}).get('obs_handoff_probe') or d.get('obs_handoff_probe')
print(json.dumps(p)[:1500] if p else 'no probe')
PY
