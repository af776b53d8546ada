#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
for f in "$@"; do
  BHW_EXTRA_FLAGS="$f" python -c "from blackman_harris_win_amd import _build; _build.build_library(force=True)" > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  python - <<PY
import torch, blackman_harris_win_amd as bhw
from blackman_harris_win_amd import binding as B
o = torch.empty(1 << 26, dtype=torch.int32, device="cuda")
pt = bhw.make_params(1, 26, 16, sin_type=B.SIN_TAYLOR, combine=B.COMBINE_VHDL, lut_size=9)
for _ in range(30): bhw.generate(pt, 0, 1 << 26, out=o)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): bhw.generate(pt, 0, 1 << 26, out=o)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)/100
print("[%s] taylor hamming 2^26/16 %.4f ms  %.0f GB/s" % ("$f", ms, 4*(1<<26)/ms/1e6))
PY
done
done
