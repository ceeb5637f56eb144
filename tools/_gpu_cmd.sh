mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r4_j_pytest.log
tail -25 gpurun_out/r4_j_pytest.log
python tools/fragment_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_j_fragment.txt
python bench.py --steps 20 --warmup 5 2>gpurun_out/r4_j_bench.err | tail -1 > gpurun_out/r4_j_bench.json
python bench.py --gpus 1 --spawn --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r4_j_bench_spawn.err | tail -1 > gpurun_out/r4_j_bench_spawn.json
python -c "
import json
for f in ('gpurun_out/r4_j_bench.json','gpurun_out/r4_j_bench_spawn.json'):
    j=json.load(open(f)); r=j['roofline']
    print(f, j['value'], j['ms_per_step'], r['kernel_ms'], r['frac'], j.get('handoff'))
"
