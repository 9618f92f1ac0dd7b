#!/bin/bash
# Run on the GPU box: SQ counters (their own pass, no trace domains) of `bench.py --method <m>`, averaged per kernel.
#   scripts/pmc_method.sh <vgicp|ndt|loam> <tag>
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
m=$1; tag=$2
OUT=$R/gpurun_out/pmc_$tag
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --method $m --steps 20 --warmup 3 --no-cpu-baseline --no-extra --windows 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_IFETCH SQ_INSTS_SMEM --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:48]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k in acc:
        n = max(cnt[k].values())
        if n < 10: continue
        print(sub, k, {c: round(v / cnt[k][c], 1) for c, v in acc[k].items()})
PY
