export TMPDIR=/tmp
ROOT=$PWD
for HOOK in "" "--debug no_qtab_row=1"; do
  (cd /tmp && timeout 200 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d /tmp/qr_"${HOOK: -1}" -o p --output-format csv -- python $ROOT/bench.py --streams 1 --steps 1 --warmup 1 --cpu-sample 0 --other off $HOOK > /dev/null 2>&1)
  python - /tmp/qr_"${HOOK: -1}"/p_counter_collection.csv "$HOOK" <<'P'
import csv, sys, collections
g = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if 'fragment_kernel' in r['Kernel_Name'] or 'settle' in r['Kernel_Name']:
        k = (r['Kernel_Name'].split('(')[0][-40:], r['Dispatch_Id'])
        g[k][r['Counter_Name']] = g[k].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
best = {}
for (name, d), v in g.items():
    if name not in best or v.get('SQ_INSTS_VALU', 0) > best[name].get('SQ_INSTS_VALU', 0): best[name] = v
for name, v in best.items(): print('hook [%s] %s SQ_INSTS_VALU %.1f M  cycles %.2f M' % (sys.argv[2], name, v['SQ_INSTS_VALU'] / 1e6, v['GRBM_GUI_ACTIVE'] / 8e6))
P
done
