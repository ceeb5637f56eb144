# scratch: the command file `gpurun -- 'bash tools/_gpu_cmd.sh'` runs on the GPU box (rewritten per call during development)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $O/r05_pytest1.txt; tail -3 $O/r05_pytest1.txt
timeout 600 python tools/numerics_report.py > $O/r05_numerics.txt 2>&1; tail -22 $O/r05_numerics.txt
timeout 600 python tools/clock_trace.py --out $O/clock_trace > $O/r05_clock_trace_plain.txt 2>&1; head -30 $O/r05_clock_trace_plain.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $O/clock_trace_kt -o kt --output-format csv -- python $R/tools/clock_trace.py --out $O/clock_trace2 > $O/r05_clock_trace_rocprof_run.txt 2>&1
cd $R
python tools/clock_trace.py --report $O/clock_trace2 --kernel-trace $O/clock_trace_kt > $O/r05_clock_trace.txt 2>&1; tail -40 $O/r05_clock_trace.txt
./tools/micro/lds_b128 > $O/r05_lds_b128.txt 2>&1; cat $O/r05_lds_b128.txt
cd /tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace -d $O/lds_pmc -o lds --output-format csv -- $R/tools/micro/lds_b128 > $O/lds_pmc.log 2>&1
python - <<'PY' > $O/r05_lds_b128_pmc.txt 2>&1
import csv, glob, collections, os
fs = glob.glob(os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/lds_pmc/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in fs:
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    print(k[:40], {c: sum(v) / len(v) for c, v in agg[k].items()})
PY
cat $O/r05_lds_b128_pmc.txt
rm -rf $O/lds_pmc
