#!/bin/bash
# A/B of compile-time variants on the GPU box: rebuild libbhw.so with each flag set, run the bench, print ms/step.
# usage: bash tools/ab_flags.sh "<flags A>" "<flags B>" ...
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
for f in "$@"; do
  BHW_EXTRA_FLAGS="$f" python -c "from blackman_harris_win_amd import _build; _build.build_library(force=True)" > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%-40s ms/step %.4f  dev_ms %.4f  parity %s' % ('[$f]', r['ms_per_step'], r['roofline']['device_ms_per_step'], r['parity_spot_check']))"
done
done
