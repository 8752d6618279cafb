set -u
mkdir -p gpurun_out/r04b
timeout 600 python -m pytest tests/test_gpu_debug_paths.py tests/test_gpu_raster_parity.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -5
for D in "" "--debug no_qpath=1"; do
  for S in 1 2; do
    echo "== streams $S $D: $(python bench.py --streams $S --steps 10 --warmup 3 --cpu-sample 0 $D 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["kernels_ms"])')"
  done
done
export TMPDIR=/tmp; R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04b/stats -o r --output-format csv -- python $R/bench.py --streams 1 --steps 5 --warmup 2 --cpu-sample 0 > $R/gpurun_out/r04b/bench_profiled.json 2>&1)
head -8 $(find gpurun_out/r04b/stats -name '*kernel_stats.csv' | head -1)
