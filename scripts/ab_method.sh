#!/bin/bash
# A/B of library variants on the VGICP or NDT line: scripts/ab_method.sh <vgicp|ndt> <out-dir> <tag> [<tag> ...]   (each tag: ab/lib<tag>.so)
m=$1; out=$2; shift; shift
mkdir -p $out
cp simpleslam_amd/lib/libpcr_hip.so $out/lib_orig.so
for rep in 1 2; do for t in "$@"; do
  cp ab/lib$t.so simpleslam_amd/lib/libpcr_hip.so
  echo "$t rep$rep $(timeout -k 10 200 python bench.py --method $m --steps 40 --warmup 5 --cpu-budget-s 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'scans/s', round(d['ms_per_step'],4), 'ms  prep', round(r['target_prep_ms'],4), 'align', round(r['align_ms'],4), 'rmse', d.get('pose_rmse_vs_cpu',{}).get('trans_m'), 'repeats', d.get('region_repeats'))")"
done; done
cp $out/lib_orig.so simpleslam_amd/lib/libpcr_hip.so
