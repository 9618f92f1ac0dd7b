#!/usr/bin/env python3
"""Timeline of ONE call out of a rocprofv3 --kernel-trace run of bench.py --method vgicp|ndt (scripts/stats_method.sh):
python scripts/trace_call.py <dir with *_kernel_trace.csv> <vgicp|ndt> [call index]   -> start offset, duration (us), queue, kernel"""
import csv, glob, sys

d, method = sys.argv[1], sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else 30
f = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True), key=lambda p: -len(open(p).read()))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
anchor, loop = ("vgicp_voxel_kernel", "vgicp_pass_pro") if method == "vgicp" else ("ndt_voxel_kernel", "ndt_pass_pro")
k = [i for i, n in enumerate(names) if anchor in n][which]
j = k
while j > 0 and loop not in names[j - 1]:
    j -= 1
e = k
while e < len(rows) - 1 and not (loop in names[e] and loop not in names[e + 1]):
    e += 1
t0 = int(rows[j]["Start_Timestamp"])
for r in rows[j:e + 1]:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  q{r['Queue_Id']:>3} {r['Kernel_Name'][:72]}")
