for ARGS in "--width 320 --height 200 --poses 8192" "--width 1280 --height 720 --poses 2048" "--width 3840 --height 2160 --poses 256" "--big" "--big --width 3840 --height 2160 --poses 256 --time-varying"; do
  python bench.py $ARGS --steps 10 --warmup 2 --cpu-sample 0 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"][:60], d["value"], d["ms_per_step"], d["config"]["kernels_ms"])'
done
