R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_bigb_gpu.py -x -q 2>&1 | tail -6 | cut -c1-250
timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -k dense 2>&1 | tail -2
B="python bench.py --no-also --no-cpu-baseline --no-stream --steps 100 --warmup 10"
for cfg in "--envs 4096 --ues 512 --bs 10" "--envs 1024 --ues 1000 --bs 10" "--envs 8192 --ues 32 --bs 64" "--envs 2048 --ues 128 --bs 64" "--envs 65536 --ues 32 --bs 40"; do echo "== $cfg"; $B $cfg 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print(r['kernel'], 'kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],3), 'env-steps/s', round(j['value']))"; done
