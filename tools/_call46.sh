set -u
timeout 600 python -m pytest tests/test_gpu_fastmath.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests -x -q -m gpu -k "debug_paths or golden or raster_parity or full_size or kat" 2>&1 | tail -3
bash tools/ab_so.sh _variants/cur.so _variants/new.so -- --other off 2>&1
