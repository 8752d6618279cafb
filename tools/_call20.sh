set -u
cp rust-doom_amd/librdoom_hip.so /tmp/_s.so; cp _variants/faddr2.so rust-doom_amd/librdoom_hip.so
timeout 600 python -m pytest tests/test_gpu_raster_parity.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -2
cp /tmp/_s.so rust-doom_amd/librdoom_hip.so
AB_ARGS="--streams 1 --poses 512 --steps 4 --warmup 1 --cpu-sample 0 --other off" bash tools/ab_cycles.sh r04s/abc shipped faddr2 2>&1 | grep -v "^round" | grep "variant\|fragment_kernel\|raster"
line() { python bench.py --cpu-sample 0 --other off --steps 10 --warmup 2 "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["config"]["kernels_ms"])'; }
for round in 1 2; do for v in shipped faddr2; do
  [ $v = shipped ] && cp /tmp/_s.so rust-doom_amd/librdoom_hip.so || cp _variants/$v.so rust-doom_amd/librdoom_hip.so
  echo "round $round [$v] $(line)"
done; done
cp /tmp/_s.so rust-doom_amd/librdoom_hip.so
