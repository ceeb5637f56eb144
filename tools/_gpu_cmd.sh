R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf $O/clock_trace2 $O/ct_kt2
timeout 600 rocprofv3 --kernel-trace -d $O/ct_kt2 -o kt --output-format csv -- python $R/tools/clock_trace.py --launches 3000 --out $O/clock_trace2 > $O/ct2_run.log 2>&1
cd $R
timeout 300 python tools/clock_trace.py --report $O/clock_trace2 --kernel-trace $O/ct_kt2 > $O/r05b_c3_clock_trace.txt 2>&1
tail -60 $O/r05b_c3_clock_trace.txt | cut -c1-200
rm -rf $O/ct_kt2
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r05_bench_line.json; python -c "
import json; j = json.load(open('$O/r05_bench_line.json')); r = j['roofline']; print(j['value'], j['ms_per_step'], r['frac'], r['kernel_ms'], r['traffic'], j['cpu_baseline']['value'], r['valu'])"
