#!/usr/bin/env python3
"""Development aid: timeline of ONE step of the caller's loop out of a rocprofv3 --kernel-trace run of scripts/seq_breakdown.py:
python scripts/trace_update.py <dir with *_kernel_trace.csv> [assembly index] [kernels after]   -> start offset, duration (us), kernel"""
import csv, glob, sys

d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 40
after = int(sys.argv[3]) if len(sys.argv) > 3 else 60
f = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True), key=lambda p: -len(open(p).read()))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
k = [i for i, n in enumerate(names) if "submap_transform" in n][which]
t0 = int(rows[k]["Start_Timestamp"])
for r in rows[max(0, k - 3):k + after]:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  {r['Kernel_Name'][:100]}")
