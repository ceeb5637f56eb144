mkdir -p gpurun_out
for pad in 0 16000 32000 48000; do echo "pad $pad"; DCOMP_FRAG_PAD_LDS=$pad python tools/fragment_bench.py 2>&1 | grep -v amdgpu | cut -c1-60; done | tee gpurun_out/r4_at_fragpad.txt
