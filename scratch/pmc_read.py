import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if sys.argv[2] in r['Kernel_Name']]
last = max(int(r['Dispatch_Id']) for r in rows)
for r in rows:
    if int(r['Dispatch_Id']) == last:
        print(r['Counter_Name'], r['Counter_Value'])
