#!/bin/bash
# A/B helper (GPU box, repo root, through gpurun): the bench lines an experiment is judged by, one compact row each, for
# every library variant named (files _variants/<name>.so; "shipped" = the library as built), two interleaved rounds
#   tools/ab_lines.sh <tag> name1 name2 ... [-- extra bench args]      -> gpurun_out/<tag>_lines.txt
# rows: workload | Mpixel/s (stream pool) | ms per step | set-up + binning / rasteriser / fragment stage (ms, single-stream pass)
TAG=${1:-ab}; shift || true
vars=(); extra=()
while [ $# -gt 0 ]; do if [ "$1" == "--" ]; then shift; extra=("$@"); break; fi; vars+=("$1"); shift; done
[ ${#vars[@]} -eq 0 ] && vars=(shipped)
OUT=gpurun_out/${TAG}_lines.txt
SETS=("" "--width 320 --height 200 --poses 8192" "--big" "--big --width 3840 --height 2160 --poses 256 --time-varying" "--width 3840 --height 2160 --poses 256" "--levels 0-8 --share 8")
[ -n "${AB_SHORT:-}" ] && SETS=("" "--width 320 --height 200 --poses 8192" "--big" "--big --width 3840 --height 2160 --poses 256 --time-varying")
cp rust-doom_amd/librdoom_hip.so /tmp/_shipped.so
echo "# $(date -u +%FT%TZ)  extra: ${extra[*]}" > $OUT
for round in 1 2; do
  for v in "${vars[@]}"; do
    if [ "$v" = shipped ]; then cp /tmp/_shipped.so rust-doom_amd/librdoom_hip.so; else cp _variants/$v.so rust-doom_amd/librdoom_hip.so; fi
    for ARGS in "${SETS[@]}"; do
      python bench.py $ARGS --steps 10 --warmup 2 --cpu-sample 0 --other off "${extra[@]}" 2>/dev/null | grep '^{' | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
k = d['config'].get('kernels_ms') or {}
sp = d.get('scaling_proxy')
print('r$round %-8s %-62s %9.1f  %7.3f ms   %.3f / %.3f / %.3f%s' % ('$v', '$ARGS' or '(default: E1M1 1080p x 1024)', d['value'] / 1e3, d['ms_per_step'], k.get('setup', 0), k.get('raster', 0), k.get('fragment', 0),
      ('   share %.3f ms, predicted %.2fx at 8' % (sp['share_ms'], sp['predicted_speedup_at_8'])) if sp else ''))
" >> $OUT
    done
  done
done
cp /tmp/_shipped.so rust-doom_amd/librdoom_hip.so
cat $OUT
