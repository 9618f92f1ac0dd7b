"""Per-dispatch PMC values of one kernel from rocprofv3 --pmc CSV output (last N dispatches)."""
import csv, glob, sys, collections, os
root, kname, last = sys.argv[1], sys.argv[2], int(sys.argv[3])
f = max(glob.glob(root + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if kname in r['Kernel_Name']]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
ids = sorted(by)[-last:]
names = sorted({c for i in ids for c in by[i]})
print('dispatch'.ljust(10) + ''.join(n[-18:].rjust(19) for n in names))
for i in ids:
    print(str(i).ljust(10) + ''.join(f"{by[i].get(n, 0):19.0f}" for n in names))
