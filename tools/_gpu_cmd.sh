mkdir -p gpurun_out
python -m pytest tests/test_fragment_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed"
python tools/fragment_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_bb_fragment.txt
