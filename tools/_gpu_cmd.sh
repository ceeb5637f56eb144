R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/fuzz_parity.py --cases 400 --seed 10101 --many-stations 0.3 --many-ues 1.0 > $O/r05_fuzz_many_ues.txt 2>&1; tail -6 $O/r05_fuzz_many_ues.txt | cut -c1-400
