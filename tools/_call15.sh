line() { python bench.py --cpu-sample 0 --other off --steps 10 --warmup 2 "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["config"]["kernels_ms"])'; }
cp rust-doom_amd/librdoom_hip.so /tmp/_orig.so
for round in 1 2; do for v in shipped rnear3 rnear5 rnear8; do
  [ $v = shipped ] && cp /tmp/_orig.so rust-doom_amd/librdoom_hip.so || cp _variants/$v.so rust-doom_amd/librdoom_hip.so
  echo "round $round [$v] default $(line) | 320x200 $(line --width 320 --height 200 --poses 8192) | big $(line --big)"
done; done
cp /tmp/_orig.so rust-doom_amd/librdoom_hip.so
