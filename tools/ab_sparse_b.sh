#!/bin/bash
# GPU box: previous library (variants/libdcomp_hip_prev.so, sparse pre-move pass for B 7..11 only) against the in-tree one
# (B 7..23) on shapes with 12 <= B <= 23.  usage: tools/ab_sparse_b.sh
cd $GRAFT_REPO_ROOT
run() {  # tag lib envs ues bs kind
  DCOMP_LIB=$2 python bench.py --no-cpu-baseline --no-also --no-stream --steps 300 --warmup 320 --envs $3 --ues $4 --bs $5 --kind $6 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('%-5s %6d x %3d x %2d %-7s kernel %.4f ms | %.3e env-steps/s | %.1f %%' % ('$1', $3, $4, $5, '$6', r['kernel_ms'], j['value'], 100 * r['frac']))"
}
PREV=$GRAFT_REPO_ROOT/deepcomp_amd/csrc/variants/libdcomp_hip_prev.so
NEW=$GRAFT_REPO_ROOT/deepcomp_amd/csrc/libdcomp_hip.so
for shape in "32768 32 16 multi" "32768 32 12 multi" "16384 32 20 multi" "8192 64 16 multi" "32768 16 16 central" "65536 32 10 multi"; do
  for rep in 1 2; do
    run prev $PREV $shape
    run new $NEW $shape
  done
done
