#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for the bench workload.
# usage: tools/profile_gpu.sh <tag> [bench args...]     outputs under gpurun_out/prof_<tag>/    (PROF_STEPS=2000: a longer kernel trace,
# whose average is the steady state rather than the cold start of the profiled process)
set -u
TAG=${1:-r1}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-stream --no-also --sustained-launches 0 $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $BENCH --steps ${PROF_STEPS:-300} --warmup 20 > $OUT/trace_bench.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/pmc_$name -o pmc --output-format csv -- $BENCH --steps 12 --warmup 3 > $OUT/pmc_$name.log 2>&1
done
python $GRAFT_REPO_ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
