mkdir -p gpurun_out
export DCOMP_BUILD_B=32
python tools/ab_lib.py run cur fakep faken --rounds 2 --only c5,c5big 2>&1 | tail -5 | tee gpurun_out/r4_p_ab.txt
