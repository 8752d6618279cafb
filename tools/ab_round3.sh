#!/bin/bash
# one-off A/B of round 3's fragment-kernel variants on one box (run from the repo root through gpurun)
set -u
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
line() { python bench.py --cpu-sample 0 --steps 10 --warmup 2 "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["kernels_ms"], "step", d["ms_per_step"], d["config"].get("debug",""))'; }
cp rust-doom_amd/librdoom_hip.so /tmp/_orig.so
for round in 1 2; do
  for v in base new nomagic; do
    cp _variants/$v.so rust-doom_amd/librdoom_hip.so
    echo "round $round [$v] $(line)"
  done
  cp _variants/new.so rust-doom_amd/librdoom_hip.so
  echo "round $round [new bw2] $(line --debug frag_bw=2)"
  echo "round $round [new no_qtab] $(line --debug no_qtab=1)"
  echo "round $round [new bw2 no_qtab] $(line --debug frag_bw=2 --debug no_qtab=1)"
  echo "round $round [new streams2] $(line --streams 2)"
done
cp /tmp/_orig.so rust-doom_amd/librdoom_hip.so
for ARGS in "--width 320 --height 200 --poses 8192" "--big"; do
  for v in base new; do cp _variants/$v.so rust-doom_amd/librdoom_hip.so; echo "[$v] $ARGS $(line $ARGS)"; done
done
cp /tmp/_orig.so rust-doom_amd/librdoom_hip.so
