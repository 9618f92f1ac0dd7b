#!/bin/bash
# Development library (environment switches live: -DPCR_DEV_SWITCHES) -> ab/libdev.so; objects in /tmp/pcr_dev (rebuilt when older than any source or header)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=/tmp/pcr_dev; mkdir -p $O $R/ab
cd $R/simpleslam_amd/csrc
newest_hdr=$(ls -t *.h ../../include/*.h | head -1)
pids=()
for f in *.hip; do
  o=$O/${f%.hip}.o
  if [ ! -f $o ] || [ $f -nt $o ] || [ $newest_hdr -nt $o ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result -Wno-inline-asm -DPCR_DEV_SWITCHES -c $f -o $o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/libdev.so $O/*.o -ldl
echo built ab/libdev.so
