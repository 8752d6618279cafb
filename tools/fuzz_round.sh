#!/bin/bash
# Fuzz over LEVELS on the current kernels (GPU box, repo root): IWADs of fresh generator seeds, builder parity + bit-exact frames,
# partly under hooks that force the rarely taken paths; extreme poses; odd window sizes.   -> gpurun_out/<tag>_seed_fuzz.txt
TAG=${1:-r06}; FIRST=${2:-12000}
OUT=gpurun_out/${TAG}_seed_fuzz.txt
{
  echo "# kernels: $(python -c 'import bench; print(bench.kernel_source_digest())')  $(date -u +%FT%TZ)  -- IWADs of other generator seeds (builder parity + bit-exact frames, 16 poses x 3 levels each), odd window sizes included; extreme poses"
  python tests/stress_other_seeds.py 160 $FIRST 16
  RDOOM_STRESS_HOOKS="no_bins=1" python tests/stress_other_seeds.py 20 $((FIRST+1000)) 16
  RDOOM_STRESS_HOOKS="bin_threads=512 settle_max=4" python tests/stress_other_seeds.py 20 $((FIRST+1100)) 16
  RDOOM_STRESS_HOOKS="bin_threads=64 no_split=1" python tests/stress_other_seeds.py 20 $((FIRST+1200)) 16
  RDOOM_STRESS_HOOKS="entry_cap=600" python tests/stress_other_seeds.py 20 $((FIRST+1300)) 16
  python tests/stress_extreme_poses.py 512 51
  python tests/stress_extreme_poses.py 512 52
  python tests/stress_parity.py 32 53 1366 768
  python tests/stress_parity.py 32 54 322 201
} > $OUT 2>&1
grep -c " ok" $OUT; grep -c -i "mismatch\|Traceback\|error" $OUT
