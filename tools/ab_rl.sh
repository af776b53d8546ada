#!/bin/bash
# A/B of the run-length kernel's compile-time knobs on BH-7 2^26 / 16-bit, model cpp (in-process, interleaved timing).
# Prebuild in the CPU container: python tools/ab_inproc.py --build-only "" "-DBHW_RL_NARROW_MAX=0"
cd "$GRAFT_REPO_ROOT"
AB_W=16 AB_MODEL=1 AB_INNER=50 AB_ROUNDS=6 python tools/ab_inproc.py "" "-DBHW_RL_NARROW_MAX=0"
