#!/bin/bash
# GPU box: A/B of timing variants built by tools/ab/ablate.py.  usage: tools/ab/ab_variants.sh "<tag> ..." [bench args]
cd $GRAFT_REPO_ROOT
TAGS=$1; shift
for t in $TAGS; do
  DCOMP_LIB=$GRAFT_REPO_ROOT/deepcomp_amd/csrc/variants/libdcomp_hip_abl0$t.so python bench.py --no-cpu-baseline --no-also --no-stream --steps 400 --warmup 50 "$@" 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('%-8s kernel %.4f ms | step %.4f ms | %.3e env-steps/s | %.1f %%' % ('$t', r['kernel_ms'], j['ms_per_step'], j['value'], 100 * r['frac']))"
done
