mkdir -p gpurun_out
timeout 800 python tools/fuzz_parity.py --cases 4000 --seed 707070 2>&1 | tail -2 | tee gpurun_out/r4_am_fuzz.txt
