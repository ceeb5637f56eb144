for us in 0 200 500 1000 3000 10000 0; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-stream --idle-before-timing-us $us 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('idle $us us: ms_per_step %.5f kernel_ms %.5f'%(d['ms_per_step'], d['roofline']['kernel_ms']))"; done
for us in 0 0; do python bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-stream 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('force-dist: ms_per_step %.5f kernel_ms %.5f between %.4f'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['between_runs_ms_total']))"; done
