#!/bin/bash
# A/B of library variants on the VGICP line: scripts/ab_vgicp.sh <out-dir> <tag> [<tag> ...]   (each tag: ab/lib<tag>.so)
out=$1; shift
mkdir -p $out
for rep in 1 2; do for t in "$@"; do
  export PCR_LIB=$(pwd)/ab/lib$t.so      # (the loader's override: the product library is not touched)
  for mode in region full; do
    flag=""; [ $mode = full ] && flag="--full-target"
    echo "$t rep$rep $mode $(timeout -k 10 200 python bench.py --method vgicp --steps 40 --warmup 5 --cpu-budget-s 2 $flag 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'scans/s', round(d['ms_per_step'],4), 'ms  prep', round(r['target_prep_ms'],4), 'align', round(r['align_ms'],4), 'rmse', d.get('pose_rmse_vs_cpu',{}).get('trans_m'))")"
  done
done; done
unset PCR_LIB
