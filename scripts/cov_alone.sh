#!/bin/bash
# Run on the GPU box with a DEV library (ab/libdev.so): the covariance search kernels alone, statistics and SQ counters.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PCR_LIB=$R/ab/libdev.so      # (the loader's override: the product library is not touched)
cd /tmp && export TMPDIR=/tmp
export PCR_COV_LPQ=4 PCR_COV_GROUP=4 PCR_COV_WAVE_BLOCKS=2048
OUT=$R/gpurun_out/cov_alone; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/scripts/cov_alone.py > $OUT/log 2>&1
tail -2 $OUT/log
python3 - $(find $OUT/stats -name '*kernel_stats.csv' | head -1) <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'cov_' in r['Name']: print(r['Name'].replace('(anonymous namespace)::','')[:40], r['Calls'], 'avg %.1f min %.1f max %.1f'%(float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/a -- python3 $R/scripts/cov_alone.py > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_IFETCH SQ_INSTS_SMEM --output-format csv -d $OUT/b -- python3 $R/scripts/cov_alone.py > $OUT/b.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k in acc:
        if 'cov_' in k: print(sub, k, {c: round(v / cnt[k][c]) for c, v in acc[k].items()})
PY
