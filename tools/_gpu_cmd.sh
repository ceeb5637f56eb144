# scratch: the command file `gpurun -- 'bash tools/_gpu_cmd.sh'` runs on the GPU box (rewritten per call during development)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
rm -rf $O/profiles_r05
bash tools/profile_all.sh r05 2>&1 | tail -14 | cut -c1-260
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_line.json 2> $O/r05_bench_line.err; tail -c 200 $O/r05_bench_line.json
python bench.py --gpus 1 --spawn --steps 20 --warmup 5 --no-cpu-baseline > $O/r05_bench_line_spawn.json 2> $O/r05_bench_line_spawn.err; tail -c 200 $O/r05_bench_line_spawn.json
bash tools/measure_configs.sh > $O/r05_measure_configs.txt 2>&1; tail -20 $O/r05_measure_configs.txt | cut -c1-200
