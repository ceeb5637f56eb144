timeout 1750 python tools/fuzz_parity.py --cases 2500 --seed 111111 2>&1 | tail -14 | cut -c1-700 | tee gpurun_out/r04_fuzz_seed111111.txt
