#!/bin/bash
# Run on the GPU box: kernel statistics of the index build alone at 1 M and 10 M points:  scripts/index_prof.sh <tag> [lib]
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; [ -n "$2" ] && export PCR_LIB=$R/ab/lib$2.so      # (the loader's override: the product library is not touched)
cd /tmp && export TMPDIR=/tmp
for n in 1000000 10000000; do
  OUT=$R/gpurun_out/index_${tag}_$n; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/scripts/index_prof.py $n > $OUT/log 2>&1
  grep "index build" $OUT/log
  python3 - $(find $OUT/stats -name '*kernel_stats.csv' | head -1) <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "grid_" in r["Name"] and ("true>" in r["Name"].split("(")[0] or "tile_kernel<8" in r["Name"] or "tile_kernel<16, 0" in r["Name"]): print('   ', r['Name'].split('(')[0][:48], r['Calls'], 'avg %.1f min %.1f max %.1f us'%(float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
