#!/bin/bash
# GPU box: PMC passes of the fused rollout kernel.  usage: tools/profile_rollout.sh <tag> E U B kind T
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/tools/bench_rollout.py $* 2000"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $CMD > $OUT/trace_bench.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/pmc_$name -o pmc --output-format csv -- $CMD > $OUT/pmc_$name.log 2>&1
done
python $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -v "elementwise\|copyBuffer\|bench line" $OUT/summary.txt | cut -c1-700
