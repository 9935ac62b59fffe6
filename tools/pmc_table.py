import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[(r['Kernel_Name'][:70], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(agg.items()):
    if max(v) > 1000:
        print(f'{c:12s} n={len(v):3d} max={max(v):14.1f} last={v[-1]:14.1f}  {k}')
