set -u
mkdir -p gpurun_out/r04g
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
( time python bench.py > gpurun_out/r04g/bench_default.json 2> gpurun_out/r04g/bench_default.err ) 2>&1 | grep real
tail -1 gpurun_out/r04g/bench_default.json | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print(d["value"], d["ms_per_step"], d["config"]["kernels_ms"], d["value_spread"])
for o in d["other_workloads"]: print(o)
print(d["roofline"])
print(d["cpu_baseline"]["value"], d["single_stream"])'
tail -3 gpurun_out/r04g/bench_default.err
