set -u
run() { python bench.py --cpu-sample 0 --steps 10 --warmup 2 --other off "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["kernels_ms"], "step", d["ms_per_step"], "single", d["single_stream"]["ms_per_step"])'; }
for W in "" "--big" "--width 320 --height 200 --poses 8192"; do
for D in "" "--debug bin_threads=128" "--debug bin_threads=64" "--streams 3"; do
echo "== $W $D: $(run $W $D)"
done; done
