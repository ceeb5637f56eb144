#!/bin/bash
# GPU box: the tracked profiles of one round in one go.  usage: tools/profile_all.sh <prefix>   (e.g. r03d)
# For each workload: tools/profile_gpu.sh / profile_rollout.sh, then summary + rocprofv3's kernel-stats table are copied to
# gpurun_out/profiles_<prefix>/ under the names profiles/ uses (<prefix>_<workload>_summary.txt, ..._kernel_stats.csv).
set -u
P=${1:-r03d}
cd $GRAFT_REPO_ROOT
DST=$GRAFT_REPO_ROOT/gpurun_out/profiles_$P
mkdir -p $DST
keep() {   # <tag>
  cp gpurun_out/prof_$1/summary.txt $DST/$1_summary.txt
  f=$(find gpurun_out/prof_$1/trace -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $DST/$1_kernel_stats.csv
  rm -rf gpurun_out/prof_$1/trace gpurun_out/prof_$1/pmc_*/        # raw traces stay on the box (gpurun_out/ is capped at 64 MiB)
}
bash tools/profile_gpu.sh ${P}_c3 > /dev/null 2>&1; keep ${P}_c3
bash tools/profile_gpu.sh ${P}_c4share --envs 32768 > /dev/null 2>&1; keep ${P}_c4share
bash tools/profile_gpu.sh ${P}_c5 --envs 4096 --ues 128 --bs 32 > /dev/null 2>&1; keep ${P}_c5
bash tools/profile_gpu.sh ${P}_c5big --envs 32768 --ues 128 --bs 32 > /dev/null 2>&1; keep ${P}_c5big
bash tools/profile_gpu.sh ${P}_c3compact --compact-step > /dev/null 2>&1; keep ${P}_c3compact
bash tools/profile_gpu.sh ${P}_c5compact --envs 4096 --ues 128 --bs 32 --compact-step > /dev/null 2>&1; keep ${P}_c5compact
bash tools/profile_gpu.sh ${P}_c5bigcompact --envs 32768 --ues 128 --bs 32 --compact-step > /dev/null 2>&1; keep ${P}_c5bigcompact
bash tools/profile_gpu.sh ${P}_central --envs 65536 --ues 10 --bs 5 --kind central > /dev/null 2>&1; keep ${P}_central
bash tools/profile_gpu.sh ${P}_big64 --envs 8192 --ues 32 --bs 64 > /dev/null 2>&1; keep ${P}_big64        # the generic kernel (33 ... 64 stations)
bash tools/profile_rollout.sh ${P}_c2roll 4096 10 5 central 100 > /dev/null 2>&1; keep ${P}_c2roll
bash tools/profile_rollout.sh ${P}_centralroll 65536 10 5 central 50 > /dev/null 2>&1; keep ${P}_centralroll
ls -la $DST
grep -h "AverageNs" $DST/*_summary.txt | grep -v "elementwise\|copyBuffer\|fill\|reset_kernel" | cut -c1-220
