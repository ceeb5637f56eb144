mkdir -p gpurun_out
./tools/micro/store_patterns 2>&1 | tee gpurun_out/r4_aj_store_patterns.txt | grep -E "^(6|3|4) "
