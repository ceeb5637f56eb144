date
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|^E " | tail -15
date
bash tools/profile_all.sh r04 2>&1 | grep AverageNs | cut -c1-200
PROF_STEPS=3000 bash tools/profile_gpu.sh r04_c3long > /dev/null 2>&1
cp gpurun_out/prof_r04_c3long/summary.txt gpurun_out/profiles_r04/r04_c3long_summary.txt
f=$(find gpurun_out/prof_r04_c3long/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/profiles_r04/r04_c3long_kernel_stats.csv
bash tools/profile_gpu.sh r04_c5compact --envs 4096 --ues 128 --bs 32 --compact-step > /dev/null 2>&1
cp gpurun_out/prof_r04_c5compact/summary.txt gpurun_out/profiles_r04/r04_c5compact_summary.txt
f=$(find gpurun_out/prof_r04_c5compact/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/profiles_r04/r04_c5compact_kernel_stats.csv
rm -rf gpurun_out/prof_r04_*/trace gpurun_out/prof_r04_*/pmc_*/
date
python tools/compact_bench.py 2>&1 | grep " x " > gpurun_out/r04_compact_step.txt; cat gpurun_out/r04_compact_step.txt
timeout 600 python tools/fuzz_parity.py --cases 400 --seed 909090 2>&1 | tail -4 | cut -c1-600 | tee gpurun_out/r04_fuzz_seed909090.txt
date
