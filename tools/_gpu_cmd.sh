python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r3_t16_pytest.log
tail -4 gpurun_out/r3_t16_pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3_t16_bench.json
python -c "
import json; j=json.load(open('gpurun_out/r3_t16_bench.json')); r=j['roofline']
print(j['value'], j['ms_per_step'], r['kernel_ms'], r['frac'], r['traffic'], r['traffic_source'][:60])
for k,v in j['also'].items():
    if isinstance(v, dict): print(k, {a:(round(b,5) if isinstance(b,float) else b) for a,b in v.items() if a in ('ms_per_step','kernel_ms','frac_of_hbm_peak','through_rollout_T50')})
"
