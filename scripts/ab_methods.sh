#!/bin/bash
# A/B of the VGICP / NDT lines on one box: scripts/ab_methods.sh <out-dir>   (region-limited preparation vs pcr_params.full_target = 1)
out=${1:-gpurun_out/ab_methods}
mkdir -p $out
for m in vgicp ndt; do
  for mode in region full; do
    flag=""; [ $mode = full ] && flag="--full-target"
    timeout -k 10 300 python bench.py --method $m --steps 40 --warmup 5 --cpu-budget-s 3 $flag > $out/${m}_$mode.json 2> $out/${m}_$mode.err || exit 1
    echo "$m $mode $(tail -1 $out/${m}_$mode.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'scans/s', round(d['ms_per_step'],4), 'ms  prep', round(r['target_prep_ms'],4), 'align', round(r['align_ms'],4), 'rmse', d.get('pose_rmse_vs_cpu',{}).get('trans_m'))")"
  done
done
