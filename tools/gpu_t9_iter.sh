#!/bin/bash
# k_tile9 iteration: GPU parity suite, then the in-process A/B of the current sources (+ timing-only floors) against the round-4 library.
# usage (GPU box, repo root): bash tools/gpu_t9_iter.sh <tag> [variant flags ...]     (variants prebuilt: tools/ab_inproc.py --build-only)
tag=${1:-t9}; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$tag.log 2>&1
  rc=$?; tail -3 gpurun_out/pytest_$tag.log
  [ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR|E )" gpurun_out/pytest_$tag.log | head -20; exit $rc; }
fi
export AB_UNITS=${AB_UNITS:-bhw_tile9.hip} AB_NOCHECK=1 AB_SPLIT=1 AB_EXTRA_LIBS=build/ab/libbhw_r4final.so AB_ROUNDS=${AB_ROUNDS:-5}
timeout -k 10 500 python tools/ab_inproc.py "$@" > gpurun_out/ab_$tag.txt 2>&1 || { tail -20 gpurun_out/ab_$tag.txt; exit 1; }
cat gpurun_out/ab_$tag.txt
