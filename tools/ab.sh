#!/bin/bash
# A/B timing on one GPU box: tools/ab.sh "ENV1=a ENV2=b" "ENV1=c" ...  (each argument = one variant's environment)
# Prints kernel times (mean over --steps) per variant, interleaved twice to expose drift.
for round in 1 2; do
  for v in "$@"; do
    out=$(env $v python bench.py --cpu-sample 0 --steps 10 --warmup 2 2>/dev/null | tail -1)
    echo "round $round [$v] $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["kernels_ms"], "step", d["ms_per_step"])')"
  done
done
