#!/bin/bash
# A/B of the fused kernel's short-launch forms (in-process, interleaved timing; AUTO picks the fused kernel at these sizes).
# Prebuild the variants in the CPU container first: python tools/ab_inproc.py --build-only "" "-DBHW_FD_SMALL_MODE=1"
cd "$GRAFT_REPO_ROOT"
for cfg in "4 20 24" "4 16 24" "7 16 32" "7 19 32" "5 18 24" "1 12 16"; do
  set -- $cfg
  echo "== BH-$1 2^$2 / $3-bit"
  AB_WIN=$1 AB_PW=$2 AB_W=$3 AB_INNER=200 AB_ROUNDS=6 python tools/ab_inproc.py "" "-DBHW_FD_SMALL_MODE=1" 2>&1 | tail -2
done
