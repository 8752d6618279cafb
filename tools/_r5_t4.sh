#!/bin/bash
mkdir -p gpurun_out/r5
rm -f gpurun_out/r5/t4_*.log
for hook in "" "--debug no_settle=1" "--debug settle_max=16" "--debug settle_max=64"; do
  echo "== 1080p $hook" >> gpurun_out/r5/t4_bench.log
  python bench.py --other off --cpu-sample 0 --steps 20 --warmup 3 $hook >> gpurun_out/r5/t4_bench.log 2>&1
done
for hook in "" "--debug no_settle=1"; do
  echo "== 320 $hook" >> gpurun_out/r5/t4_bench.log
  python bench.py --other off --cpu-sample 0 --steps 10 --warmup 3 --width 320 --height 200 --poses 8192 --streams 3 $hook >> gpurun_out/r5/t4_bench.log 2>&1
  echo "== big $hook" >> gpurun_out/r5/t4_bench.log
  python bench.py --other off --cpu-sample 0 --steps 10 --warmup 3 --big $hook >> gpurun_out/r5/t4_bench.log 2>&1
  echo "== 4k $hook" >> gpurun_out/r5/t4_bench.log
  python bench.py --other off --cpu-sample 0 --steps 10 --warmup 3 --width 3840 --height 2160 --poses 256 $hook >> gpurun_out/r5/t4_bench.log 2>&1
done
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5/t4_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r5/t4_pytest.log
