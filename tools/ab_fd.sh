#!/bin/bash
# A/B of the fused kernel's compile-time knobs on short windows (in-process, interleaved timing; AUTO picks the fused kernel).
# Prebuild the variants in the CPU container first: python tools/ab_inproc.py --build-only "" "-DBHW_FD_LOCKSTEP_MAX=0"
cd "$GRAFT_REPO_ROOT"
for cfg in "4 20 24" "4 16 24" "7 16 32" "7 20 32" "5 18 24"; do
  set -- $cfg
  echo "== BH-$1 2^$2 / $3-bit"
  AB_WIN=$1 AB_PW=$2 AB_W=$3 AB_INNER=200 AB_ROUNDS=6 python tools/ab_inproc.py "" "-DBHW_FD_LOCKSTEP_MAX=0" 2>&1 | tail -2
done
