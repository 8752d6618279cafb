#!/bin/bash
# A/B of library variants on one box: tools/ab_generic.sh name1 name2 ... (files _variants/<name>.so); two interleaved rounds
# of the default bench, then the 320x200 and the large-level lines for each (run from the repo root through gpurun)
set -u
line() { python bench.py --cpu-sample 0 --steps 10 --warmup 2 "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["config"]["kernels_ms"], "step", d["ms_per_step"])'; }
cp rust-doom_amd/librdoom_hip.so /tmp/_orig.so
for round in 1 2; do for v in "$@"; do cp _variants/$v.so rust-doom_amd/librdoom_hip.so; echo "round $round [$v] $(line)"; done; done
for v in "$@"; do cp _variants/$v.so rust-doom_amd/librdoom_hip.so
  echo "[$v] 320x200 $(line --width 320 --height 200 --poses 8192)"; echo "[$v] big $(line --big)"
  python -m pytest tests/test_gpu_raster_parity.py tests/test_gpu_full_size.py tests/test_gpu_debug_paths.py -q -m gpu 2>&1 | tail -1
done
cp /tmp/_orig.so rust-doom_amd/librdoom_hip.so
