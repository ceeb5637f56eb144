date
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
date
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|^E " | tail -8
date
s=$(date +%s); python bench.py > gpurun_out/bench_default.json 2>/dev/null; e=$(date +%s); echo "default bench.py: $((e-s)) s"; tail -c 150 gpurun_out/bench_default.json; echo
s=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2>/dev/null; e=$(date +%s); echo "driver form: $((e-s)) s"
