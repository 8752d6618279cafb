#!/bin/bash
# A/B timing of library variants on one GPU box: tools/ab_so.sh _variants/a.so _variants/b.so ... [-- bench args]
# Each variant is copied over rust-doom_amd/librdoom_hip.so in turn (two interleaved rounds to expose drift).
vars=(); args=()
while [ $# -gt 0 ]; do if [ "$1" == "--" ]; then shift; args=("$@"); break; fi; vars+=("$1"); shift; done
cp rust-doom_amd/librdoom_hip.so /tmp/_orig.so
for round in 1 2; do
  for v in "${vars[@]}"; do
    cp "$v" rust-doom_amd/librdoom_hip.so
    out=$(python bench.py --cpu-sample 0 --steps 10 --warmup 2 "${args[@]}" 2>/dev/null | tail -1)
    echo "round $round [$v] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["kernels_ms"], "step", d["ms_per_step"])')"
  done
done
cp /tmp/_orig.so rust-doom_amd/librdoom_hip.so
