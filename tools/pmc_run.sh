#!/bin/bash
# PMC passes over a short bench run (rocprofv3, counters in their own runs: --pmc + --kernel-trace only).
# usage (on the GPU box, from the repo root): tools/pmc_run.sh <out_subdir> [bench args...]
# Writes gpurun_out/<out_subdir>/pass<N>/..._counter_collection.csv ; summarise with tools/pmc_summary.py.
set -u
OUT=gpurun_out/${1:-pmc}; shift || true
ARGS=${@:---poses 256 --steps 1 --warmup 1 --cpu-sample 0}
export TMPDIR=/tmp
mkdir -p $OUT
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $CTRS --kernel-trace -d $OUT/pass$i -o p --output-format csv -- python bench.py $ARGS > $OUT/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
find $OUT -name "*counter_collection.csv" | head
