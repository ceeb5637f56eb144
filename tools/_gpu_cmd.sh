mkdir -p gpurun_out
export DCOMP_BUILD_B=10
python tools/ab_lib.py run rev0 rev1 rev1p --rounds 2 --only c3,c4share,c3rf 2>&1 | tail -6 | tee gpurun_out/r4_r_ab.txt
