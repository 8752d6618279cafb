#!/bin/bash
mkdir -p gpurun_out/r5
python bench.py --other off --cpu-sample 0 --steps 20 --warmup 3 > gpurun_out/r5/t3_bench.log 2>&1
python bench.py --other off --cpu-sample 0 --steps 10 --warmup 3 --width 320 --height 200 --poses 8192 --streams 3 > gpurun_out/r5/t3_bench_320.log 2>&1
python bench.py --other off --cpu-sample 0 --steps 10 --warmup 3 --big > gpurun_out/r5/t3_bench_big.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5/t3_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r5/t3_pytest.log
