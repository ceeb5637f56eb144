# scratch: the command file `gpurun -- 'bash tools/_gpu_cmd.sh'` runs on the GPU box (rewritten per call during development)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_bigb_gpu.py -q -x 2>&1 | tail -40 > $O/r05_bigb.txt; tail -40 $O/r05_bigb.txt | cut -c1-300
timeout 600 python -m pytest tests/test_parity_gpu.py -q -x -k "dense" 2>&1 | tail -15 | cut -c1-300
