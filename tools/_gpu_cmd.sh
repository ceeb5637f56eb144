mkdir -p gpurun_out
python -m pytest tests/test_parity_gpu.py -m gpu -q -k "capped_occupancy" 2>&1 | grep -E "passed|failed"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4_av_bench.json
python -c "
import json
j=json.load(open('gpurun_out/r4_av_bench.json')); a=j['also']
print({k:(round(v['kernel_ms'],4),round(v['frac_of_hbm_peak'],3)) for k,v in a['config5_per_gpu_share_of_32768x128x32'].items()}, round(a['config5_strong_32768x128x32']['frac_of_hbm_peak_per_gpu'],3))"
