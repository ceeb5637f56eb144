mkdir -p gpurun_out
export DCOMP_BUILD_B=5
python tools/ab_lib.py run ch8 ch16 ch32 --rounds 2 --only c2roll,central10x5roll,c2policy 2>&1 | tail -6 | tee gpurun_out/r4_z_ab.txt
