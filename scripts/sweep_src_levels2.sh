#!/bin/bash
# Development sweep (needs ab/libdev.so = a `make DEV=1` build): a FINER first level for the scan's own search (cell0 in voxels)
out=${1:-gpurun_out/sweep_src}
mkdir -p $out
export PCR_LIB=$(pwd)/ab/libdev.so      # (the loader's override: the product library is not touched)
run() {      # cell0 ratio levels
  r=$(PCR_COV_CELL0=$1 PCR_COV_RATIO=$2 PCR_COV_LEVELS=$3 timeout -k 10 200 python bench.py --method vgicp --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['target_prep_ms'],4), round(d['roofline']['align_ms'],4))")
  echo "cell0 $1 ratio $2 levels $3 -> $r"
}
run 1 6 2
run 0.5 4 3
run 0.5 3 3
run 0.5 6 3
run 0.5 12 2
run 0.25 4 3
run 0.7 4 3
run 1 6 2
unset PCR_LIB
