#!/bin/bash
# Run on the GPU box: kernel-trace statistics of `bench.py --method <m> [flags]`, printed per kernel with the calls PER SCAN.
#   scripts/stats_method.sh <vgicp|ndt|loam> <tag> [bench flags]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
m=$1; tag=$2; shift 2
OUT=$R/gpurun_out/stats_$tag
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=40
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --method $m --steps $STEPS --warmup 5 --no-cpu-baseline --no-extra --windows 1 "$@" > $OUT/stats.log 2>&1
f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    if float(r["TotalDurationNs"]) / tot < 0.004: continue
    print(f'{r["Name"].split("(")[0][:64]:64s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us  {100 * float(r["TotalDurationNs"]) / tot:5.1f}%')
PY
grep -h '"metric"' $OUT/stats.log | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("value", round(d["value"],1), "ms/step", round(d["ms_per_step"],4))'
