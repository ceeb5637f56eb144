# scratch: the command file `gpurun -- 'bash tools/_gpu_cmd.sh'` runs on the GPU box (rewritten per call during development)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -14 > $O/r05_gpu_pytest_tail2.txt; grep -E "passed|failed|FAILED|Error" $O/r05_gpu_pytest_tail2.txt | head -8
one() { if [ -n "$1" ]; then export DCOMP_LIB=$R/deepcomp_amd/csrc/variants/libdcomp_hip_$1.so; else unset DCOMP_LIB; fi; python bench.py --no-cpu-baseline --no-also --no-stream --steps 400 --warmup 50 $2 $3 $4 $5 $6 $7 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('%-8s kernel %.4f ms | steady %.4f | step %.4f ms | %.1f %%' % ('${1:-tree}', r['kernel_ms'], r['steady_state']['kernel_ms'], j['ms_per_step'], 100 * r['frac']))"; unset DCOMP_LIB; }
for i in 1 2; do one abl0; one ""; done 2>&1 | tee $O/r05_ab_tree.txt
echo "== tree: config 5 share, config 5 whole"; one "" --envs 4096 --ues 128 --bs 32; one "" --envs 32768 --ues 128 --bs 32
timeout 700 python tools/fuzz_parity.py --cases 1500 --seed 959595 --many-stations 0.15 > $O/r05_fuzz_seed959595.txt 2>&1; tail -2 $O/r05_fuzz_seed959595.txt | cut -c1-300
