mkdir -p gpurun_out
for kb in 8 4 2 12 16; do echo "cap ${kb} KB"; DCOMP_FRAG_CAP_KB=$kb python tools/fragment_bench.py 2>&1 | grep -v amdgpu.ids | head -4; done | tee gpurun_out/r4_t_frag.txt
echo noverify; DCOMP_FRAG_NOVERIFY=1 python tools/fragment_bench.py 2>&1 | grep "pack" | head -4 | tee -a gpurun_out/r4_t_frag.txt
