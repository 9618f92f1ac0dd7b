#!/bin/bash
# Run on the GPU box: kernel-trace statistics of a short bench run, printed as a compact table (iteration aid).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/quick_stats
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > $OUT/stats.log 2>&1
f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f'{r["Name"].split("(")[0][:60]:60s} {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us  {r["Percentage"]:>6s}%')
PY
grep -h '"metric"' $OUT/stats.log | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("value", d["value"], "ms/step", d["ms_per_step"])'
