#!/bin/bash
# A/B of LOAM library variants on one box: scripts/ab_loam.sh <out-dir> <tag> [<tag> ...]   (each tag: ab/lib<tag>.so)
# Per variant: the per-launch timeline (scripts/iter_profile.py) and the headline bench line, twice round robin.
out=$1; shift
mkdir -p $out
for rep in 1 2; do
for t in "$@"; do
  export PCR_LIB=$(pwd)/ab/lib$t.so      # (the loader's override: the product library is not touched)
  [ $rep = 1 ] && { timeout -k 10 200 python scripts/iter_profile.py > $out/ip_$t.log 2>&1 || echo "iter_profile failed for $t"; }
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra > $out/bk_$t$rep.json 2> $out/bk_$t$rep.err || { echo "bench failed for $t"; continue; }
  echo "$t rep$rep $(tail -1 $out/bk_$t$rep.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(round(d["value"],1), "scans/s  launch", round(r["avg_launch_us"],2), "us  index", round(r["index_build_us"],1), "us  kept", round(d["index_kept"]["value"],1), " rmse", d.get("pose_rmse_vs_cpu"))')"
done; done
unset PCR_LIB
for t in "$@"; do echo "== $t"; grep -E "^[0-9] prologue|^searches" $out/ip_$t.log; done
