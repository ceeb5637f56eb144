#!/bin/bash
# GPU box: A/B of timing variants on the fused rollout.  usage: tools/ab/ab_rollout.sh "<tag> ..." E U B kind T
cd $GRAFT_REPO_ROOT
TAGS=$1; shift
for t in $TAGS; do
  echo -n "$t  "; DCOMP_LIB=$GRAFT_REPO_ROOT/deepcomp_amd/csrc/variants/libdcomp_hip_abl0$t.so python tools/bench_rollout.py "$@" 8000 | cut -c1-75
done
