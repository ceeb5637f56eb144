# scratch: the command file `gpurun -- 'bash tools/_gpu_cmd.sh'` runs on the GPU box (rewritten per call during development)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 1200 python -m pytest tests/test_c_host_gpu.py tests/test_bigb_gpu.py tests/test_adapters_gpu.py -x -q 2>&1 | tail -12 | cut -c1-300
timeout 900 python -m pytest tests/test_handoff_gpu.py -x -q --durations=4 2>&1 | tail -9 | cut -c1-200
bash tools/profile_gpu.sh r05_c5compact --envs 4096 --ues 128 --bs 32 --compact-step > /dev/null 2>&1
mkdir -p $O/profiles_r05; cp $O/prof_r05_c5compact/summary.txt $O/profiles_r05/r05_c5compact_summary.txt; f=$(find $O/prof_r05_c5compact/trace -name '*kernel_stats.csv' | head -1); cp $f $O/profiles_r05/r05_c5compact_kernel_stats.csv; rm -rf $O/prof_r05_c5compact/trace $O/prof_r05_c5compact/pmc_*/
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_line.json 2> $O/r05_bench_line.err; tail -c 300 $O/r05_bench_line.json
