# scratch: the command file `gpurun -- 'bash tools/_gpu_cmd.sh'` runs on the GPU box (rewritten per call during development)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
for m in nopin pin pin_restore torch_nopin torch_pin; do python tools/_cpu_diag.py $m 2>&1 | tail -1; done
timeout 2400 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -40 > $O/r05_pytest2.txt; tail -14 $O/r05_pytest2.txt
