#!/bin/bash
OUT=$PWD/gpurun_out/r5; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/final_pytest.log 2>&1
echo "rc=$?" >> $OUT/final_pytest.log
SEEDS=60 bash tools/stress_round.sh r05w > $OUT/final_stress_tail.log 2>&1
