#!/bin/bash
# GPU box: kernel-trace stats + FETCH/WRITE PMC passes of the heuristic policy kernel in the loop with dcomp_step (config 3).
# usage: tools/profile_policy.sh <tag>     outputs under gpurun_out/prof_<tag>/
set -u
TAG=${1:-policy}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/tools/bench_policy.py"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $CMD > $OUT/trace_bench.log 2>&1
for grp in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/pmc_$grp -o pmc --output-format csv -- $CMD > $OUT/pmc_$grp.log 2>&1
done
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -8 {}' > $OUT/summary.txt
python - $OUT >> $OUT/summary.txt <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
for grp in ('FETCH_SIZE', 'WRITE_SIZE'):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f'{out}/pmc_{grp}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:60]
            acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
    for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:4]:
        print(f'{grp} {k}: {v / n:.1f} KB per launch over {n} launches (gfx950: FETCH_SIZE counts half, see MI355X_MICROARCH.md)')
P
tail -4 $OUT/trace_bench.log >> $OUT/summary.txt
cat $OUT/summary.txt
