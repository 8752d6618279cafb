set -u
timeout 900 python -m pytest tests/test_gpu_debug_paths.py tests/test_gpu_raster_parity.py tests/test_gpu_full_size.py tests/test_big_level.py tests/test_gpu_stress_slice.py -x -q -m gpu 2>&1 | tail -5
export TMPDIR=/tmp; R=$PWD
cp rust-doom_amd/librdoom_hip.so /tmp/_ship.so
for v in shipped qocc4 qocc6 qt8 qt2; do
  [ $v = shipped ] && cp /tmp/_ship.so rust-doom_amd/librdoom_hip.so || cp _variants/$v.so rust-doom_amd/librdoom_hip.so
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04d/$v -o r --output-format csv -- python $R/bench.py --streams 1 --steps 4 --warmup 1 --cpu-sample 0 > /dev/null 2>&1)
  echo "== $v"; grep -h "fragment_quadrant\|fragment_kernel" $(find gpurun_out/r04d/$v -name '*kernel_stats.csv') | sed 's/(.*)",/",/' | cut -c1-120
done
cp _variants/fstats.so rust-doom_amd/librdoom_hip.so
python bench.py --streams 1 --steps 1 --warmup 0 --poses 256 --cpu-sample 0 2>&1 | grep -a "frag stats" | tail -2
cp /tmp/_ship.so rust-doom_amd/librdoom_hip.so
for S in 1 2; do echo "== streams $S: $(python bench.py --streams $S --steps 10 --warmup 3 --cpu-sample 0 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["kernels_ms"], d["config"]["alpha_leak_fixup_pixels_per_step"])')"; done
