#!/bin/bash
# Counter passes aimed at what limits fragment_kernel / raster_wave_kernel (issue slots, texture addresser, vector L1).
# NOT in the list: the TA_* / TD_* and the TCP_*_STALL_CYCLES / TCP_GATE_EN* counters -- every pass that asked for them hung
# rocprofv3 on this pool's boxes until its timeout (three passes, 15 GPU-minutes).  Each pass runs under `timeout 120`.
# usage (GPU box, repo root): tools/pmc_frag.sh <out_subdir> [bench args...]; summarise with tools/pmc_summary.py
set -u
OUT=gpurun_out/${1:-pmcf}; shift || true
ARGS=${@:---streams 1 --poses 256 --steps 1 --warmup 1 --cpu-sample 0}
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_available.txt 2>&1
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
            "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_TAGCONFLICT_STALL_CYCLES_sum" \
            "SQ_IFETCH SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $CTRS --kernel-trace -d $OUT/pass$i -o p --output-format csv -- python bench.py $ARGS > $OUT/pass$i.log 2>&1
  echo "pass $i rc=$? ($CTRS)"
done
find $OUT -name "*counter_collection.csv" | head -20
