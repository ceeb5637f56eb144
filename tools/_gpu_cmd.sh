mkdir -p gpurun_out
python -m pytest tests/test_handoff_gpu.py tests/test_rollout_gpu.py tests/test_parity_gpu.py -m gpu -q -x -k "handoff or driver_form or single_rank or spawns or live_reseed or event_schedule or is_fused or checkpoint or two_ranks" 2>&1 | tail -15 > gpurun_out/r4_a_pytest.log
tail -15 gpurun_out/r4_a_pytest.log
python bench.py --steps 20 --warmup 5 2>gpurun_out/r4_a_bench.err | tail -1 > gpurun_out/r4_a_bench.json
python bench.py --gpus 1 --spawn --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r4_a_bench_spawn.err | tail -1 > gpurun_out/r4_a_bench_spawn.json
python -c "
import json
for f in ('gpurun_out/r4_a_bench.json','gpurun_out/r4_a_bench_spawn.json'):
    j=json.load(open(f)); r=j['roofline']
    print(f, j['value'], j['ms_per_step'], r['kernel_ms'], r['frac'], j.get('handoff'))
    for k,v in j['also'].items():
        if isinstance(v, dict): print(' ', k, {a:(round(b,5) if isinstance(b,float) else b) for a,b in v.items() if a in ('value','ms_per_step','kernel_ms','frac_of_hbm_peak','frac_of_hbm_peak_per_gpu') or a.startswith('N')})
"
