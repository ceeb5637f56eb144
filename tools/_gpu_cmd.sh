# scratch: the command file `gpurun -- 'bash tools/_gpu_cmd.sh'` runs on the GPU box (rewritten per call during development)
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|^E " | tail -8
python bench.py --gpus 1 --steps 20 --warmup 5 | tail -c 400
