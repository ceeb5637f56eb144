cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r3_t11_pytest.log
bash tools/profile_gpu.sh r03b_c5 --envs 4096 --ues 128 --bs 32 > /dev/null 2>&1
bash tools/profile_gpu.sh r03b_c3 > /dev/null 2>&1
bash tools/profile_rollout.sh r03b_c2roll 4096 10 5 central 100 > /dev/null 2>&1
bash tools/profile_gpu.sh r03b_central --envs 65536 --ues 10 --bs 5 --kind central > /dev/null 2>&1
bash tools/profile_gpu.sh r03b_c4share --envs 32768 > /dev/null 2>&1
for t in c5 c3 c2roll central c4share; do cp gpurun_out/prof_r03b_$t/summary.txt gpurun_out/r03b_${t}_summary.txt; f=$(find gpurun_out/prof_r03b_$t/trace -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r03b_${t}_kernel_stats.csv; rm -rf gpurun_out/prof_r03b_$t; done
tail -3 gpurun_out/r3_t11_pytest.log
timeout 900 python tools/fuzz_parity.py --cases 3000 --seed 20260930 > gpurun_out/r3_fuzz1.log 2>&1
tail -4 gpurun_out/r3_fuzz1.log
