#!/bin/bash
# A/B of the run-length kernel's compile-time knobs on BH-7 2^26 / 16-bit, model cpp (in-process, interleaved timing)
cd "$GRAFT_REPO_ROOT"
AB_W=16 AB_MODEL=1 AB_INNER=50 AB_ROUNDS=6 python tools/ab_inproc.py "" "-DBHW_RL_DIRECT_STORE=1" "-DBHW_RL_BLOCK=64" "-DBHW_RL_BLOCK=256"
