# scratch: the command file `gpurun -- 'bash tools/_gpu_cmd.sh'` runs on the GPU box (rewritten per call during development)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_bigb_gpu.py -q -x 2>&1 | tail -3
timeout 600 python -m pytest tests/test_parity_gpu.py -q -x -k "dense" 2>&1 | tail -2
B="python bench.py --no-also --no-cpu-baseline --no-stream --steps 200 --warmup 20"
for cfg in "--envs 8192 --ues 32 --bs 64" "--envs 8192 --ues 32 --bs 33" "--envs 65536 --ues 32 --bs 40" "--envs 2048 --ues 128 --bs 64" "--envs 65536 --ues 10 --bs 40 --kind central"; do
  echo "== $cfg"; $B $cfg 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print(r['kernel'], 'kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],3), 'env-steps/s', round(j['value']))"
done
echo "== generic at 8192x32x32 and 65536x32x10"
DCOMP_FORCE_BIG=1 $B --envs 8192 --ues 32 --bs 32 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print(r['kernel'], 'kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],3))"
DCOMP_FORCE_BIG=1 $B 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print(r['kernel'], 'kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],3))"
