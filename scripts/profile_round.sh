#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace statistics of the bench command and the HBM traffic
# counters of the dominant kernel, each PMC group in its own pass (MI355X_MICROARCH.md, "HBM" / "rocprofv3 PMC").
# Output under gpurun_out/prof_<tag>; the summaries are then copied into profiles/ by scripts/collect_profiles.py.
set -e
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (--no-twin: the counter passes see the dispatches of ONE leg -- the hinted one the line's figures are of -- not an average over it and the stateless twin)
ARGS="$R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra --windows 1 --no-twin"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $ARGS > $OUT/pmc_l2.log 2>&1
grep -h '"metric"' $OUT/stats.log | tail -1 > $OUT/bench_under_profiler.json || true
# the two secondary configurations (BASELINE configs[2] and configs[4]): kernel statistics and the HBM traffic counters (own passes)
for m in vgicp ndt; do
  MARGS="$R/bench.py --method $m --steps 40 --warmup 5 --no-cpu-baseline --windows 1 --no-twin"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$m -- python3 $MARGS > $OUT/stats_$m.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$m -- python3 $MARGS > $OUT/pmc_fetch_$m.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$m -- python3 $MARGS > $OUT/pmc_write_$m.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq_$m -- python3 $MARGS > $OUT/pmc_sq_$m.log 2>&1
done
# the caller's loop (extra.sequence): kernel statistics of one LOAM drive (voxel filter, sub-map assembly, target builds, iterations)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_sequence -- python3 $R/scripts/seq_breakdown.py loam > $OUT/stats_sequence.log 2>&1
echo done
