#!/bin/bash
# Runs on the GPU box: the DESIGN.md section-4 table (kernel ms by HIP events, env-steps/s, achieved GB/s, fraction of 8 TB/s).
cd $GRAFT_REPO_ROOT
row() { python bench.py --no-cpu-baseline --no-also --no-stream "$@" 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('%-70s kernel %.4f ms | step %.4f ms | %.3e env-steps/s | %.0f GB/s | %.1f %%' % (j['config']['workload'][:70], r['kernel_ms'], j['ms_per_step'], j['value'], r['achieved'], 100 * r['frac']))"; }
row
row --sharing resource-fair
row --envs 4096 --ues 10 --bs 5 --kind central --steps 2000
row --envs 65536 --ues 10 --bs 5 --kind central
row --envs 4096 --ues 128 --bs 32 --steps 400
row --envs 32768 --ues 128 --bs 32 --steps 200
row --envs 262144 --ues 32 --bs 10 --steps 200
# fused rollout (dcomp_rollout_ex: T steps per launch, every step's outputs written), secondary figures
for shape in "4096 10 5 central 100" "4096 32 10 multi 20" "1024 10 5 central 100"; do
  python tools/bench_rollout.py $shape 8000 --launches | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('fused rollout $shape: %.3f us/step | %.3e env-steps/s | one launch per step: %.3e env-steps/s' % (j['ms_per_step'] * 1e3, j['value'], j.get('one_launch_per_step_env_steps_per_s', 0)))"
done
