#!/bin/bash
# A/B of DEV-build environment switches on one box: scripts/ab_env.sh <vgicp|ndt|loam> "<ENV=.. ENV=..>" "<...>" ...   (library: ab/libdev.so; "-" = no switch)
m=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PCR_LIB=$R/ab/libdev.so      # (the loader's override: the product library is not touched)
for rep in 1 2; do for cfg in "$@"; do
  envs=""; [ "$cfg" != "-" ] && envs="$cfg"
  echo "[$cfg] rep$rep $(env $envs timeout -k 10 200 python bench.py --method $m --steps 40 --warmup 5 --cpu-budget-s 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'scans/s', round(d['ms_per_step'],4), 'ms  prep', round(r.get('target_prep_ms',0),4), 'align', round(r.get('align_ms',0),4), 'rmse', d.get('pose_rmse_vs_cpu',{}).get('trans_m'))")"
done; done
