python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err; tail -c 200 gpurun_out/r04_bench_line.json; echo
python bench.py --gpus 1 --steps 20 --warmup 5 --compact-step --no-also --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_line_compact_step.json
python bench.py --gpus 1 --spawn --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_line_spawn.json
ls -la gpurun_out/r04_bench_line*.json
