R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for t in r4 new r4 new; do
  export PCR_LIB=$R/ab/lib$t.so
  OUT=$R/gpurun_out/idx10_$t; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/scripts/index_prof.py 10000000 > $OUT/log 2>&1
  echo "== $t $(grep 'index build' $OUT/log)"
  python3 - $(find $OUT/stats -name '*kernel_stats.csv' | head -1) <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"])>5: print('   ', r['Name'].split('(')[0][:60], r['Calls'], 'avg %.1f min %.1f max %.1f us'%(float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
