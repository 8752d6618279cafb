python -m pytest tests/test_golden.py tests/test_gpu_raster_parity.py -m gpu -x -q 2>&1 | tail -2
for a in "" "--big" "--width 320 --height 200 --poses 8192"; do cd /tmp; rm -rf /tmp/prof; rocprofv3 --kernel-trace --stats -d /tmp/prof -o r --output-format csv -- python $GRAFT_REPO_ROOT/bench.py $a --steps 3 --warmup 1 --cpu-sample 0 > /tmp/b.json 2>/dev/null; cd $GRAFT_REPO_ROOT; python - <<PY
import csv,glob,json
d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1]); print("== $a", d["value"], d["config"]["kernels_ms"])
for f in glob.glob("/tmp/prof/**/r_kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print("  ", r["Name"].replace("rdoom_dev::(anonymous namespace)::","")[:40], r["Calls"], round(float(r["AverageNs"])/1000,1))
PY
done
