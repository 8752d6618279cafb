set -u
timeout 900 python -m pytest tests/test_gpu_raster_parity.py tests/test_gpu_debug_paths.py tests/test_gpu_full_size.py tests/test_big_level.py tests/test_gpu_stress_slice.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -4
python tools/raster_stats.py 2>&1 | grep "rdoom stats" | tail -1
python tools/raster_stats.py --width 320 --height 200 --poses 2048 2>&1 | grep "rdoom stats" | tail -1
python tools/raster_stats.py --big 2>&1 | grep "rdoom stats" | tail -1
for S in 1 2; do echo "== streams $S: $(python bench.py --streams $S --steps 10 --warmup 3 --cpu-sample 0 --other off 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["kernels_ms"])')"; done
