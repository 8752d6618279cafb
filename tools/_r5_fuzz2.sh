#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
{
echo "# kernels: $(python -c 'import bench; print(bench.kernel_source_digest())')  $(date -u +%FT%TZ)  -- round 5, second batch on the shipped sources"
python tests/stress_other_seeds.py 240 13000 16
python tests/stress_extreme_poses.py 512 52
RDOOM_STRESS_HOOKS="no_pair=1" python tests/stress_other_seeds.py 30 14000 16
RDOOM_STRESS_HOOKS="settle_max=4 bin_threads=512" python tests/stress_other_seeds.py 30 14500 16
RDOOM_STRESS_HOOKS="no_settle=1" python tests/stress_big_level.py
python tests/stress_big_level.py
python tests/stress_parity.py 32 63 1921 1082
python tests/stress_parity.py 32 64 1284 724
} > $OUT/r05w_seed_fuzz2.txt 2>&1
grep -c " ok" $OUT/r05w_seed_fuzz2.txt; grep -ci "MISMATCH\|Traceback" $OUT/r05w_seed_fuzz2.txt; tail -2 $OUT/r05w_seed_fuzz2.txt
