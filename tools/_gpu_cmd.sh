mkdir -p gpurun_out
DCOMP_BUILD_B=10 python tools/ab_lib.py run m0 m1 --rounds 2 --only c3,c4share,c3rf,central32x10 2>&1 | tail -6 | tee gpurun_out/r4_ah_ab.txt
DCOMP_BUILD_B=5 python tools/ab_lib.py run n0 n1 --rounds 2 --only central10x5,c2roll,central10x5roll 2>&1 | tail -5 | tee -a gpurun_out/r4_ah_ab.txt
