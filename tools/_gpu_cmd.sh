mkdir -p gpurun_out
DCOMP_BUILD_B=5 python tools/ab_lib.py run r0 r1 --rounds 2 --only central10x5roll,c2roll,central10x5 2>&1 | tail -5
