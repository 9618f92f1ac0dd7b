"""Condense gpurun_out/prof_<tag> (scripts/profile_round.sh <tag>) into profiles/: the kernel statistics table, the
per-kernel PMC means, and profiles/loam_iterate_pmc.json (HBM bytes per launch of the dominant kernel,
corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE are in KiB... see below)."""
import csv, glob, json, os, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def kname(n):
    """kernel name without its argument list (the kernels of an anonymous namespace keep their own name)"""
    return n.replace("(anonymous namespace)::", "").split("(")[0]


def newest(pattern):
    fs = glob.glob(os.path.join(src, pattern), recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None


f = newest("stats/**/*kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline\n")
        o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows:
            o.write(",".join([kname(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]]) + "\n")

for m in ("vgicp", "ndt"):
    f = newest(f"stats_{m}/**/*kernel_stats.csv")
    if f:
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(dst, f"{tag}_{m}_kernel_stats.csv"), "w") as o:
            o.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --method {m} --steps 40 --warmup 5 --no-cpu-baseline\n")
            o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
            for r in rows:
                o.write(",".join([kname(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]]) + "\n")

f = newest("stats_sequence/**/*kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(dst, f"{tag}_sequence_kernel_stats.csv"), "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats -- python3 scripts/seq_breakdown.py loam   (two drives of 64 scans: the caller's loop of extra.sequence, LOAM)\n")
        o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows:
            o.write(",".join([kname(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]]) + "\n")


def means(pattern):
    f = newest(pattern)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            acc[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in d.items()} for k, d in acc.items()}


allm = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_l2"):
    for k, d in means(f"{sub}/**/*counter_collection.csv").items():
        allm.setdefault(k, {}).update(d)
with open(os.path.join(dst, f"{tag}_pmc_means.csv"), "w") as o:
    o.write("# mean counter value per dispatch; one rocprofv3 --pmc pass per counter group (separate runs)\n")
    o.write("Kernel,Counter,MeanPerDispatch,Dispatches\n")
    for k in sorted(allm):
        for c in sorted(allm[k]):
            o.write(f"{k},{c},{allm[k][c][0]:.1f},{allm[k][c][1]}\n")

it = next((v for k, v in allm.items() if "loam_iterate_kernel" in k), {})      # templated: "void pcr::loam_iterate_kernel<10, 1>"
if "FETCH_SIZE" in it and "WRITE_SIZE" in it:
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE
    # counts 128-B requests as 64 B for wide coalesced streams -> the guide doubles it for streaming reads.  This
    # kernel's reads are 16-B gathers and 128-B cache entries, an uncalibrated pattern: both figures are recorded.
    fetch_kib, write_kib = it["FETCH_SIZE"][0], it["WRITE_SIZE"][0]
    out = {
        "kernel": "loam_iterate_kernel",
        "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib,
        "hbm_bytes_per_launch": (2 * fetch_kib + write_kib) * 1024,
        "hbm_bytes_per_launch_uncorrected": (fetch_kib + write_kib) * 1024,
        "note": "mean over all launches of the bench run (full-search and cache-hit iterations); FETCH_SIZE doubled per the gfx950 correction",
    }
    json.dump(out, open(os.path.join(dst, "loam_iterate_pmc.json"), "w"), indent=1)
    json.dump(out, open(os.path.join(dst, f"{tag}_loam_iterate_pmc.json"), "w"), indent=1)
    print(out)
# configs[2] / configs[4]: HBM bytes per scan of everything but the optimiser's passes (= target preparation, plus the scan's own two index
# levels and covariances for VGICP), from the per-dispatch counters: sum over the kernels of (mean bytes per dispatch x dispatches per scan)
for m in ("vgicp", "ndt"):
    fm, wm = means(f"pmc_fetch_{m}/**/*counter_collection.csv"), means(f"pmc_write_{m}/**/*counter_collection.csv")
    if not fm or not wm:
        continue
    # one call = one launch of the kernel that builds the voxels (the optimiser's initial state no longer has a launch of its own in NDT)
    scans = next((v["FETCH_SIZE"][1] for k, v in fm.items() if "ctl_store" in k and m == "vgicp"), 0) or next((v["FETCH_SIZE"][1] for k, v in fm.items() if "_voxel_kernel" in k), 0)
    rows, prep, total = [], 0.0, 0.0
    for k in sorted(set(fm) | set(wm)):
        f_kib, n = fm.get(k, {}).get("FETCH_SIZE", (0.0, 0))
        w_kib, n2 = wm.get(k, {}).get("WRITE_SIZE", (0.0, 0))
        per_scan = (2 * f_kib * n + w_kib * n2) * 1024 / max(1, scans)      # FETCH_SIZE doubled: the gfx950 correction of the guide
        is_pass = "pass_pro" in k or "ctl_store" in k or "copyBuffer" in k
        rows.append({"kernel": k, "dispatches_per_scan": n / max(1, scans), "fetch_kib": f_kib, "write_kib": w_kib, "hbm_bytes_per_scan": per_scan, "target_preparation": not is_pass})
        total += per_scan
        prep += 0.0 if is_pass else per_scan
    extra = {}
    sq = means(f"pmc_sq_{m}/**/*counter_collection.csv")
    with open(os.path.join(dst, f"{tag}_{m}_pmc_sq_means.csv"), "w") as o:
        o.write("# mean SQ counter value per dispatch (own rocprofv3 --pmc pass)\nKernel,Counter,MeanPerDispatch,Dispatches\n")
        for k in sorted(sq):
            for c in sorted(sq[k]):
                o.write(f"{k},{c},{sq[k][c][0]:.1f},{sq[k][c][1]}\n")
    if m == "ndt":
        pk = next((v for k, v in sq.items() if "ndt_pass_pro_kernel" in k), None)
        if pk and "SQ_ACTIVE_INST_VALU" in pk and "SQ_WAVE_CYCLES" in pk:
            extra["valu_active_frac"] = pk["SQ_ACTIVE_INST_VALU"][0] / pk["SQ_WAVE_CYCLES"][0]
            extra["valu_active_note"] = "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES of ndt_pass_pro_kernel, mean over its dispatches"
    if m == "vgicp":
        for k in rows:
            if "vgicp_cov_kernel<false>" in k["kernel"]:
                extra["hbm_bytes_per_launch_cov_target"] = (2 * k["fetch_kib"] + k["write_kib"]) * 1024
    json.dump({"method": m, "scans": scans, "leg": "the hinted leg only: bench.py --no-twin, the stateless twin and its full builds do not run in the counter passes",
               "hbm_bytes_per_scan_preparation": prep, "hbm_bytes_per_scan_total": total, **extra, "kernels": rows,
               "note": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch (rocprofv3 --pmc, one counter per pass), FETCH doubled per the gfx950 correction; "
                       "preparation = every kernel but the optimiser's passes"}, open(os.path.join(dst, f"{tag}_{m}_pmc.json"), "w"), indent=1)
    print(m, "HBM bytes per scan: preparation", prep, "total", total)
b = os.path.join(src, "bench_under_profiler.json")
if os.path.exists(b):
    open(os.path.join(dst, f"{tag}_bench_under_profiler.json"), "w").write(open(b).read())
print("profiles written to", dst)
