#!/bin/bash
# long randomized parity runs on the current kernels (run from the repo root through gpurun); the log goes to gpurun_out/<tag>_stress.txt
TAG=${1:-r03}
{
  echo "# kernels: $(python -c 'import bench; print(bench.kernel_source_digest())')  $(date -u +%FT%TZ)"
  for seed in 31 32 33; do python tests/stress_parity.py 64 $seed; done
  python tests/stress_parity.py 24 34 1920 1080
  python tests/stress_parity.py 64 35 320 200
  python tests/stress_extreme_poses.py 256 41
  python tests/stress_extreme_poses.py 256 42
  python tests/stress_big_level.py
} > gpurun_out/${TAG}_stress.txt 2>&1
tail -3 gpurun_out/${TAG}_stress.txt
grep -c " ok" gpurun_out/${TAG}_stress.txt; grep -c MISMATCH gpurun_out/${TAG}_stress.txt
