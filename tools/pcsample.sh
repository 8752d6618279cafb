#!/bin/bash
# PC sampling of the hot kernels (rocprofv3 --pc-sampling-beta-enabled): tools/pcsample.sh <out_subdir> [bench args...]
# Tries the stochastic (hardware) method first, then host_trap; every attempt under its own timeout.  The per-instruction
# table is made by tools/pcsample_summary.py from the csv the successful attempt leaves.
set -u
OUT=$PWD/gpurun_out/${1:-pcs}; shift || true
ARGS=${@:---streams 1 --poses 256 --steps 2 --warmup 1 --cpu-sample 0}
export TMPDIR=/tmp
mkdir -p $OUT
ROOT=$PWD
cd /tmp
timeout 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576 \
    --kernel-trace -d $OUT/stochastic -o p --output-format csv -- python $ROOT/bench.py $ARGS > $OUT/stochastic.log 2>&1
echo "stochastic rc=$?"; tail -3 $OUT/stochastic.log
timeout 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1 \
    --kernel-trace -d $OUT/host_trap -o p --output-format csv -- python $ROOT/bench.py $ARGS > $OUT/host_trap.log 2>&1
echo "host_trap rc=$?"; tail -3 $OUT/host_trap.log
find $OUT -type f | head -20
for f in $(find $OUT -name '*pc_sampling*.csv'); do echo "== $f"; head -5 $f; wc -l $f; done
