mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r4_al_bench.json
python -c "
import json
j=json.load(open('gpurun_out/r4_al_bench.json')); r=j['roofline']
print(j['value'], j['ms_per_step'], r['kernel_ms'], r['frac'], r['traffic'], r['traffic_source'][:50], r['steady_state']['frac'])
a=j['also']; print(a['config5_share_4096x128x32_multi']['frac_of_hbm_peak'], a['config4_share_32768x32x10_multi']['frac_of_hbm_peak'], a['config5_strong_32768x128x32']['frac_of_hbm_peak_per_gpu'], a['central_65536x10x5']['frac_of_hbm_peak'])"
