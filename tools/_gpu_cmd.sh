mkdir -p gpurun_out
python -m pytest tests/test_handoff_gpu.py -m gpu -q -k "torch_distributed_run" 2>&1 | tail -15
