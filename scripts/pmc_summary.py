"""Summarise rocprofv3 --pmc CSV output: mean counter value per dispatch for each kernel."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v):5d} mean={sum(v)/len(v):16.1f}")
