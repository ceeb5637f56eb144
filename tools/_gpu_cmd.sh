mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r4_ba_bench.json
python -c "
import json
j=json.load(open('gpurun_out/r4_ba_bench.json')); r=j['roofline']; a=j['also']
print(j['value'], j['ms_per_step'], r['kernel_ms'], r['frac'], r['traffic'], r['steady_state']['frac'])
print({k:(round(v['kernel_ms'],4),round(v['frac_of_hbm_peak'],3)) for k,v in a['config5_per_gpu_share_of_32768x128x32'].items()})
print({k:(round(v['kernel_ms'],4),round(v['frac_of_hbm_peak'],3)) for k,v in a['config4_per_gpu_share_of_262144x32x10'].items()})
print(round(a['config5_strong_32768x128x32']['frac_of_hbm_peak_per_gpu'],3), round(a['config4_strong_262144x32x10']['frac_of_hbm_peak_per_gpu'],3), round(a['central_65536x10x5']['frac_of_hbm_peak'],3))"
