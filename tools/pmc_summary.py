"""Summarise rocprofv3 --pmc counter_collection.csv per kernel: sum of each counter over dispatches."""
import csv, sys, collections
for path in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0][-40:]
            agg[k][row['Counter_Name']] += float(row['Counter_Value'])
            n[(k, row['Counter_Name'])] += 1
    for k, cs in agg.items():
        print(k)
        for c, v in sorted(cs.items()):
            print('   %-24s %16.0f  (%d dispatches)' % (c, v, n[(k, c)]))
