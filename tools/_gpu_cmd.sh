cd $GRAFT_REPO_ROOT
timeout 500 python -m pytest tests/test_parity_gpu.py -q -k "crowd or dense" 2>&1 | tail -4
