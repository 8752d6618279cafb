#!/bin/bash
# A/B of library variants in the CYCLE domain (one box): tools/ab_cycles.sh <out_subdir> name1 name2 ...  (files _variants/<name>.so;
# "shipped" = the library as built).  For every variant, two interleaved rounds of ONE rocprofv3 pass
#   --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace
# over bench.py --streams 1 (every kernel alone on the GPU): the counter file carries each dispatch's start / end stamps, so
# duration, cycles, effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and instruction counts come from the SAME launches.
# A millisecond comparison alone cannot tell "the instruction count is exhausted" from "the chip clocked down"
# (MI355X_MICROARCH.md, DVFS give-back).  Summary: python tools/ab_cycles_summary.py gpurun_out/<out_subdir>
set -u
OUT=$PWD/gpurun_out/${1:-abc}; shift
ARGS=${AB_ARGS:---streams 1 --poses 512 --steps 4 --warmup 1 --cpu-sample 0}
export TMPDIR=/tmp
mkdir -p $OUT
ROOT=$PWD
cp rust-doom_amd/librdoom_hip.so /tmp/_shipped.so
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = shipped ]; then cp /tmp/_shipped.so rust-doom_amd/librdoom_hip.so; else cp _variants/$v.so rust-doom_amd/librdoom_hip.so; fi
    (cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES \
        --kernel-trace -d $OUT/$v.r$round -o p --output-format csv -- python $ROOT/bench.py $ARGS > $OUT/$v.r$round.log 2>&1)
    echo "round $round [$v] rc=$? $(grep -o '"value": [0-9.]*' $OUT/$v.r$round.log | head -1) $(grep -o '"kernels_ms": {[^}]*}' $OUT/$v.r$round.log | head -1)"
  done
done
cp /tmp/_shipped.so rust-doom_amd/librdoom_hip.so
python tools/ab_cycles_summary.py $OUT | tee $OUT/summary.txt
