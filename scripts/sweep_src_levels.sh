#!/bin/bash
# Development sweep (needs ab/libdev.so = a `make DEV=1` build): cell size and ratio of the scan's own search levels in VGICP.
out=${1:-gpurun_out/sweep_src}
mkdir -p $out
cp simpleslam_amd/lib/libpcr_hip.so $out/lib_orig.so
cp ab/libdev.so simpleslam_amd/lib/libpcr_hip.so
for c0 in 1 2 3; do for ratio in 3 4 6; do for lv in 2; do
  r=$(PCR_COV_CELL0=$c0 PCR_COV_RATIO=$ratio PCR_COV_LEVELS=$lv timeout -k 10 200 python bench.py --method vgicp --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['target_prep_ms'],4), round(d['roofline']['align_ms'],4))")
  echo "cell0 $c0 ratio $ratio levels $lv -> $r"
done; done; done
for c0 in 2 4; do
  r=$(PCR_COV_CELL0=$c0 PCR_COV_LEVELS=1 timeout -k 10 200 python bench.py --method vgicp --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['target_prep_ms'],4), round(d['roofline']['align_ms'],4))")
  echo "cell0 $c0 levels 1 -> $r"
done
cp $out/lib_orig.so simpleslam_amd/lib/libpcr_hip.so
