#!/bin/bash
# A/B of DEV-build environment switches, one repetition each, alternating: scripts/ab_env_quick.sh <vgicp|ndt|loam> "<ENV=..>" ...   ("-" = no switch)
m=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PCR_LIB=$R/ab/libdev.so
for rep in 1 2; do for cfg in "$@"; do
  envs=""; [ "$cfg" != "-" ] && envs="$cfg"
  echo "[$cfg] rep$rep $(env $envs timeout -k 10 200 python bench.py --method $m --steps 40 --warmup 5 --no-cpu-baseline --windows 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), 'scans/s', round(d['ms_per_step'],4), 'ms  prep', round(r.get('target_prep_ms',0),4), 'align', round(r.get('align_ms',0),4))")"
done; done
