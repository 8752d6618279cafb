#!/bin/bash
# per-level timing of the pose sweep (BASELINE config 4's levels): tools/levels.sh [poses]
P=${1:-256}
for L in 0 1 2 3 4 5 6 7 8; do
  out=$(python bench.py --cpu-sample 0 --steps 3 --warmup 1 --poses $P --level $L 2>/dev/null | tail -1)
  echo "level $L $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); c=d["config"]; print(c["kernels_ms"], "tris/pose", c["visible_triangles_per_pose"], "fixups", c["alpha_leak_fixup_pixels_per_step"], "Mpix/s", d["value"])')"
done
