python -m pytest tests/test_gpu_debug_paths.py tests/test_gpu_raster_parity.py -x -q 2>&1 | tail -3
for r in 1 2; do for v in "" "--debug keep_vis=1"; do python bench.py $v --steps 10 --warmup 3 --cpu-sample 0 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['value'], d['ms_per_step'], d['config']['kernels_ms'], d.get('single_stream',{}).get('ms_per_step'))"; done; done
for v in "" "--debug keep_vis=1"; do python bench.py $v --width 320 --height 200 --poses 8192 --steps 10 --warmup 3 --cpu-sample 0 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('320x200 [$v]', d['value'], d['ms_per_step'], d['config']['kernels_ms'])"; python bench.py $v --big --steps 10 --warmup 3 --cpu-sample 0 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('big [$v]', d['value'], d['ms_per_step'], d['config']['kernels_ms'])"; done
