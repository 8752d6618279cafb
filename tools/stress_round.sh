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
  # the same slices through the whole-quadrant fragment kernel (off by default: the qpath hook)
  RDOOM_STRESS_HOOKS="qpath=1" python tests/stress_parity.py 48 36
  RDOOM_STRESS_HOOKS="qpath=1" python tests/stress_parity.py 16 37 1920 1080
  RDOOM_STRESS_HOOKS="qpath=1" python tests/stress_extreme_poses.py 128 43
  RDOOM_STRESS_HOOKS="qpath=1" python tests/stress_big_level.py
  python tests/stress_other_seeds.py ${SEEDS:-60} 7000 16
  RDOOM_STRESS_HOOKS="qpath=1" python tests/stress_other_seeds.py 30 8000 16
} > gpurun_out/${TAG}_stress.txt 2>&1
tail -3 gpurun_out/${TAG}_stress.txt
grep -c " ok" gpurun_out/${TAG}_stress.txt; grep -c MISMATCH gpurun_out/${TAG}_stress.txt
