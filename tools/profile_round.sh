#!/bin/bash
# Round profile on the GPU box (run from the repo root through gpurun):  tools/profile_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats over the default `python bench.py` command      -> gpurun_out/<tag>/stats
#   2. PMC passes (counters in their own runs: --pmc + --kernel-trace only)              -> gpurun_out/<tag>/pmc*
#   3. the un-profiled bench lines: default (BASELINE config 3), other frame sizes, the level sweep (config 4 on one
#      GPU), the MAP29-class line (config 5), two ranks wrapped onto this one GPU (weak and strong)
# Then locally:  python tools/profile_collect.py <tag> <round>   copies the summaries into profiles/.
set -u
TAG=${1:-prof}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
# (a) --streams 1: every kernel alone on the GPU, one launch per step -- the per-kernel accounting (rNN_kernel_stats.csv)
rocprofv3 --kernel-trace --stats -d $OUT/stats -o r --output-format csv -- python $ROOT/bench.py --streams 1 --other off > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
echo "stats (--streams 1) rc=$?"
# (b) the default command as it is (sub-batches on the stream pool, then its single-stream pass): launches are sub-batches,
#     those of the timed region overlap (rNN_kernel_stats_default.csv + the kernel trace for tools/profile_collect.py)
rocprofv3 --kernel-trace --stats -d $OUT/stats_default -o r --output-format csv -- python $ROOT/bench.py --other off > $OUT/bench_profiled_default.json 2> $OUT/bench_profiled_default.err
echo "stats (default) rc=$?"
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --kernel-trace -d $OUT/pmc$i -o p --output-format csv -- python $ROOT/bench.py --streams 1 --steps 1 --warmup 1 --cpu-sample 0 --other off > $OUT/pmc$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
cd $ROOT
python bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
tail -1 $OUT/bench_plain.json
: > $OUT/bench_other.jsonl
for ARGS in "--width 3840 --height 2160 --poses 256" "--width 1280 --height 720 --poses 2048" "--width 320 --height 200 --poses 8192" \
            "--big" "--big --width 3840 --height 2160 --poses 256 --time-varying" "--levels 0-8 --share 8" \
            "--streams 1" "--streams 2" "--width 1366 --height 768 --poses 2048" "--gpus 2" "--gpus 2 --scaling strong" \
            "--gpus 2 --launcher threads --scaling strong"; do
  python bench.py $ARGS --steps 10 --warmup 2 --cpu-sample 0 --other off 2>/dev/null | grep '^{' | tail -1 >> $OUT/bench_other.jsonl
done
wc -l $OUT/bench_other.jsonl
# 4. issue-slot / occupancy counters of the hot kernels on the final device sources
bash tools/pmc_frag.sh $TAG/issue > $OUT/issue.log 2>&1
find $OUT/issue -name '*counter_collection.csv' | wc -l
# 5. A/B lines of the round's hooks on the same box (equivalent paths, same images): settle_kernel off / other list limits
: > $OUT/bench_hooks.jsonl
for ARGS in "--debug no_settle=1" "--debug settle_max=16" "--debug settle_max=64" "--width 3840 --height 2160 --poses 256 --debug no_settle=1" \
            "--big --debug no_settle=1" "--width 320 --height 200 --poses 8192 --debug no_settle=1" "--levels 0-8 --debug no_settle=1"; do
  python bench.py $ARGS --steps 10 --warmup 2 --cpu-sample 0 --other off 2>/dev/null | grep '^{' | tail -1 >> $OUT/bench_hooks.jsonl
done
wc -l $OUT/bench_hooks.jsonl
