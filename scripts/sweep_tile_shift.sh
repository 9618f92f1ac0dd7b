#!/bin/bash
# Development sweep (needs ab/libdev.so = a `make DEV=1` build): tile size of every index build of a method's bench line.
m=${1:-vgicp}; out=${2:-gpurun_out/sweep_tile}
mkdir -p $out
export PCR_LIB=$(pwd)/ab/libdev.so      # (the loader's override: the product library is not touched)
for ts in default 7 8 9 10 11 12 13; do
  if [ $ts = default ]; then unset PCR_TILE_SHIFT; else export PCR_TILE_SHIFT=$ts; fi
  r=$(timeout -k 10 200 python bench.py --method $m --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['target_prep_ms'],4), round(d['roofline']['align_ms'],4))")
  echo "$m tile shift $ts -> $r"
done
unset PCR_TILE_SHIFT
unset PCR_LIB
