"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel: MEAN of each counter per dispatch."""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
meta = {}
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][-40:]
            agg[k][row['Counter_Name']] += float(row['Counter_Value'])
            n[(k, row['Counter_Name'])] += 1
            meta[k] = (row['VGPR_Count'], row['Accum_VGPR_Count'], row['SGPR_Count'], row['LDS_Block_Size'], row['Grid_Size'])
for k, cs in agg.items():
    print('%s   vgpr=%s agpr=%s sgpr=%s lds=%s grid=%s' % ((k,) + meta[k]))
    for c, v in sorted(cs.items()):
        print('   %-28s %18.0f  (mean of %d dispatches)' % (c, v / n[(k, c)], n[(k, c)]))
