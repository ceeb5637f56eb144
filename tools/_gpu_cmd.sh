mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r4_ae_pytest.log
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r4_ae_bench.json
python -c "
import json
j=json.load(open('gpurun_out/r4_ae_bench.json')); r=j['roofline']
print(j['value'], j['ms_per_step'], r['kernel_ms'], r['frac'], r['traffic'], r['steady_state']['frac'])"
