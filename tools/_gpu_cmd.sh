R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 780 python tools/fuzz_parity.py --cases 4000 --seed 60606 --many-stations 0.15 --many-ues 0.03 > $O/r05_fuzz_final_library.txt 2>&1; tail -3 $O/r05_fuzz_final_library.txt | cut -c1-300
