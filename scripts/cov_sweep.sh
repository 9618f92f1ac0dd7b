#!/bin/bash
# Run on the GPU box with a DEV library (ab/libdev.so): sweep of the covariance search's launch shape on the VGICP line.
#   scripts/cov_sweep.sh "<lpq> <group> <wave_blocks>" ...      -> per configuration: parity (cov_debug), kernel averages (rocprofv3), scans/s
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PCR_LIB=$R/ab/libdev.so      # (the loader's override: the product library is not touched)
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  set -- $cfg
  export PCR_COV_LPQ=$1 PCR_COV_GROUP=$2 PCR_COV_WAVE_BLOCKS=$3
  OUT=$R/gpurun_out/covsweep_$1_$2_$3
  rm -rf $OUT && mkdir -p $OUT
  bad=$(cd $R && timeout -k 10 100 python scripts/cov_debug.py 2>&1 | grep -E "^bad" | head -1)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --method vgicp --steps 30 --warmup 5 --no-cpu-baseline --no-extra --windows 1 > $OUT/stats.log 2>&1
  f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$cfg" "$bad" $OUT/stats.log <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
g = lambda s: next((float(r["AverageNs"]) / 1e3 for r in rows if s in r["Name"]), float("nan"))
line = [l for l in open(sys.argv[4]) if '"metric"' in l]
ms = json.loads(line[-1])["ms_per_step"] if line else float("nan")
print(f'cfg {sys.argv[2]:12s} ring1 {g("cov_ring1"):7.1f} wave {g("cov_wave"):7.1f} nbr {g("cov_from_nbr"):6.1f} cov<false> {g("vgicp_cov_kernel<false>"):7.1f} us | profiled {ms:.4f} ms | {sys.argv[3]}')
PY
done
