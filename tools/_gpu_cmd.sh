set -x
for x in 0 1 0 1; do
  echo "=== DCOMP_HEUR_XCD=$x"
  DCOMP_HEUR_XCD=$x python tools/bench_policy.py --envs 65536 --ues 32 --bs 10 2>&1 | grep -E "kernel|alone"
done
for x in 0 1; do
  echo "=== big DCOMP_HEUR_XCD=$x"
  DCOMP_HEUR_XCD=$x python tools/bench_policy.py --envs 8192 --ues 128 --bs 32 2>&1 | grep -E "kernel|alone"
done
DCOMP_HEUR_XCD=1 python -m pytest tests/test_adapters_gpu.py tests/test_policy_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
