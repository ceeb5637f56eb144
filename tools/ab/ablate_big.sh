#!/bin/bash
# Timing-only ablation and A/B switches of the generic kernel (dcomp_big.h: DCOMP_BIG_ABL bits, DCOMP_BIG_NT / _EARLY / _WUNROLL / _PUNROLL).
#   here:        tools/ab/ablate_big.sh build "<tag>=<flags>" ...    e.g.  build "abl4=-DDCOMP_BIG_ABL=4" "nt=-DDCOMP_BIG_NT=1" "base="
#                (compiles dcomp_big.hip + dcomp_api.hip with the flags, links them with the product's per-station-count objects into
#                 deepcomp_amd/csrc/variants/libdcomp_hip_big<tag>.so; the variants travel to the GPU box with the snapshot)
#   GPU box:     tools/ab/ablate_big.sh run "<tag> ..." [bench args]   (DCOMP_BIG_ABL variants compute WRONG results: timing only)
REPO=$(cd "$(dirname "$0")/../.." && pwd); C=$REPO/deepcomp_amd/csrc; V=$C/variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast-honor-pragmas --offload-compress -Wall -Wno-unused-function"
if [ "$1" = build ]; then
  shift; mkdir -p $V /tmp/ablate_big
  for spec in "$@"; do
    tag=${spec%%=*}; fl=${spec#*=}
    hipcc $F $fl -c $C/dcomp_big.hip -o /tmp/ablate_big/big_$tag.o && hipcc $F $fl -c $C/dcomp_api.hip -o /tmp/ablate_big/api_$tag.o &&
      hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libdcomp_hip_big$tag.so $C/build/dcomp_inst_b*.o /tmp/ablate_big/api_$tag.o /tmp/ablate_big/big_$tag.o -lpthread && echo "built $tag"
  done
else
  shift; TAGS=$1; shift
  ARGS=${@:---envs 8192 --ues 32 --bs 64}
  cd $REPO
  for t in $TAGS; do
    DCOMP_LIB=$V/libdcomp_hip_big$t.so python bench.py --no-cpu-baseline --no-also --no-stream --no-check --steps 300 --warmup 30 $ARGS 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('%-12s kernel %.4f ms | step %.4f ms | %.1f %%' % ('$t', r['kernel_ms'], j['ms_per_step'], 100 * r['frac']))"
  done
fi
