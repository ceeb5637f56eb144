mkdir -p gpurun_out/profiles_r04
bash tools/profile_gpu.sh r04_c5big --envs 32768 --ues 128 --bs 32 > /dev/null 2>&1
cp gpurun_out/prof_r04_c5big/summary.txt gpurun_out/profiles_r04/r04_c5big_summary.txt
f=$(find gpurun_out/prof_r04_c5big/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/profiles_r04/r04_c5big_kernel_stats.csv
rm -rf gpurun_out/prof_r04_c5big/trace gpurun_out/prof_r04_c5big/pmc_*/
grep "step_kernel_wide" gpurun_out/profiles_r04/r04_c5big_summary.txt | head -1 | cut -c1-230
