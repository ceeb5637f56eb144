R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd $R
python -c "from deepcomp_amd import build; print('up_to_date', build.up_to_date())"
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -14 > $O/r05_gpu_pytest_tail5.txt; grep -E "passed|failed|FAILED|Error" $O/r05_gpu_pytest_tail5.txt | head -8
timeout 500 python tools/fuzz_parity.py --cases 400 --seed 50505 --many-stations 0.5 --many-ues 0.3 > $O/r05_fuzz_generic_final.txt 2>&1; tail -2 $O/r05_fuzz_generic_final.txt | cut -c1-300
P=r05b
DST=$O/profiles_$P; mkdir -p $DST
bash tools/profile_gpu.sh ${P}_big64 --envs 8192 --ues 32 --bs 64 > /dev/null 2>&1
cp gpurun_out/prof_${P}_big64/summary.txt $DST/${P}_big64_summary.txt
f=$(find gpurun_out/prof_${P}_big64/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $DST/${P}_big64_kernel_stats.csv
rm -rf gpurun_out/prof_${P}_big64/trace gpurun_out/prof_${P}_big64/pmc_*/
for s in "--envs 8192 --ues 32 --bs 64" "--envs 8192 --ues 32 --bs 64 --sharing resource-fair" "--envs 65536 --ues 32 --bs 40" "--envs 65536 --ues 10 --bs 40 --kind central" "--envs 2048 --ues 128 --bs 64" "--envs 4096 --ues 512 --bs 10" "--envs 1024 --ues 1000 --bs 10"; do
  for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-also --no-stream --steps 300 --warmup 30 $s 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('%-56s %-28s kernel %.4f ms | step %.4f ms | %.3e env-steps/s | frac %.3f' % ('$s', r.get('kernel','')[:28], r['kernel_ms'], j['ms_per_step'], j['value'], r['frac']))"
  done
done 2>&1 | tee $O/r05_generic_shapes.txt
DCOMP_FORCE_BIG=1 python bench.py --no-cpu-baseline --no-also --no-stream --steps 300 --warmup 30 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('FORCE_BIG 65536x32x10  kernel %.4f ms | step %.4f ms | frac %.3f' % (r['kernel_ms'], j['ms_per_step'], r['frac']))" | tee -a $O/r05_generic_shapes.txt
