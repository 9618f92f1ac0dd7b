#!/bin/bash
# Run on the GPU box: A/B of the index build under development switches (DEV library ab/libdev.so):
#   scripts/r5_index_ab.sh "<ENV=.. ENV=..>" ...     ("-" = no switch; PCR_BIN_SPLIT=0: tiles never cut, PCR_TILE_PER_CU=0: a block per bin)
R=${GRAFT_REPO_ROOT:-$(pwd)}
i=0
for cfg in "$@"; do
  i=$((i+1))
  echo "== [$cfg]"
  if [ "$cfg" = "-" ]; then bash $R/scripts/index_prof.sh ab$i dev; else env $cfg bash $R/scripts/index_prof.sh ab$i dev; fi
done
