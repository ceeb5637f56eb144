mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r4_as_pytest.log
python __graft_entry__.py smoke 2>&1 | grep "smoke ok"
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r4_as_bench.json
python -c "
import json
j=json.load(open('gpurun_out/r4_as_bench.json')); r=j['roofline']
print(j['value'], j['ms_per_step'], r['kernel_ms'], r['frac'], r['traffic'], r['steady_state']['frac'], j['cpu_baseline']['value'])"
