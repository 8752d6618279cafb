set -u
timeout 900 python -m pytest tests/test_gpu_debug_paths.py -x -q -m gpu -k "split_lists_tile_limit" 2>&1 | tail -3
SEEDS=200 python tests/stress_other_seeds.py 200 9000 16 > gpurun_out/r04_seed_fuzz.txt 2>&1; tail -2 gpurun_out/r04_seed_fuzz.txt; grep -c " ok" gpurun_out/r04_seed_fuzz.txt; grep -c MISMATCH gpurun_out/r04_seed_fuzz.txt
