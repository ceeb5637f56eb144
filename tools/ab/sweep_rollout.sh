#!/bin/bash
# GPU box: fused rollout vs one launch per step over batch sizes (finds the crossover of dcomp_create's heuristic).
cd $GRAFT_REPO_ROOT
for shape in "10 5 central" "32 10 multi" "10 5 multi"; do
 for E in 1024 4096 8192 16384 32768 65536; do
  set -- $shape
  f=$(DCOMP_FUSE_MAX_WAVES=100000000 python tools/bench_rollout.py $E $1 $2 $3 20 2000 | python -c "import sys,json; print('%.3f' % (json.loads(sys.stdin.read())['ms_per_step']*1e3))")
  l=$(DCOMP_FUSE_MAX_WAVES=0 python tools/bench_rollout.py $E $1 $2 $3 20 2000 | python -c "import sys,json; print('%.3f' % (json.loads(sys.stdin.read())['ms_per_step']*1e3))")
  echo "$shape E=$E  fused $f us/step   launches $l us/step"
 done
done
