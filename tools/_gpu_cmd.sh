mkdir -p gpurun_out
export DCOMP_BUILD_B=32
V=$GRAFT_REPO_ROOT/deepcomp_amd/csrc/variants
DCOMP_LIB=$V/libdcomp_hip_sp1.so python tools/check_wide.py 2>&1 | tail -2
python tools/ab_lib.py run sp0 sp1 sp2 --rounds 2 --only c5,c5big 2>&1 | tail -6 | tee gpurun_out/r4_i_ab.txt
