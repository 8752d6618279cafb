#!/bin/bash
OUT=$PWD/gpurun_out/r5; mkdir -p $OUT
{
for ARGS in "--other off" "--other off --streams 1" "--other off --levels 0-8 --poses 128" "--other off --big --streams 1" "--other off --width 320 --height 200 --poses 8192 --streams 1" "--other off --width 3840 --height 2160 --poses 256 --streams 1"; do
  echo "== $ARGS"
  bash tools/ab_so.sh _variants/chunksinner.so _variants/chunksouter.so -- $ARGS
done
} > $OUT/t16_ab.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/t16_pytest.log 2>&1
echo "rc=$?" >> $OUT/t16_pytest.log
