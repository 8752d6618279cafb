#!/bin/bash
# Round profile on the GPU box (run from the repo root through gpurun):
#   1. rocprofv3 --kernel-trace --stats over the default `python bench.py` command  -> gpurun_out/<tag>/stats
#   2. PMC passes (counters only, separate runs) for HBM traffic of the fragment kernel -> gpurun_out/<tag>/pmc*
# Then locally:  python tools/profile_collect.py <tag> <round>   copies the summaries into profiles/.
set -u
TAG=${1:-prof}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o r --output-format csv -- python $ROOT/bench.py > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
echo "stats rc=$?"
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --kernel-trace -d $OUT/pmc$i -o p --output-format csv -- python $ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 > $OUT/pmc$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
cd $ROOT
python bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
tail -1 $OUT/bench_plain.json
