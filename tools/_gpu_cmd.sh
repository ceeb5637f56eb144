# scratch: the command file `gpurun -- 'bash tools/_gpu_cmd.sh'` runs on the GPU box (rewritten per call during development)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > $O/r05_gpu_pytest_tail.txt; tail -4 $O/r05_gpu_pytest_tail.txt
bash tools/profile_all.sh r05 2>&1 | tail -16 | cut -c1-260
timeout 900 python tools/fuzz_parity.py --cases 1500 --seed 737373 --many-stations 0.5 > $O/r05_fuzz_seed737373.txt 2>&1; tail -2 $O/r05_fuzz_seed737373.txt | cut -c1-300
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_line.json 2> $O/r05_bench_line.err; tail -c 700 $O/r05_bench_line.json
python bench.py --gpus 1 --spawn --steps 20 --warmup 5 --no-cpu-baseline > $O/r05_bench_line_spawn.json 2> $O/r05_bench_line_spawn.err; tail -c 300 $O/r05_bench_line_spawn.json
