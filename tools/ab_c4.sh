#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
for f in "$@"; do
  BHW_EXTRA_FLAGS="$f" python -c "from blackman_harris_win_amd import _build; _build.build_library(force=True)" > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  python - <<PY
import torch, blackman_harris_win_amd as bhw
p4 = bhw.make_params(4, 16, 24)
o4 = torch.empty((1024, 1 << 16), dtype=torch.int32, device="cuda")
for _ in range(20): bhw.generate_batched(p4, 1024, out=o4)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): bhw.generate_batched(p4, 1024, out=o4)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)/100
print("[%s] C4 %.4f ms  %.0f GB/s" % ("$f", ms, 4*(1<<26)/ms/1e6))
PY
done
done
