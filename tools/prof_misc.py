#!/usr/bin/env python3
"""Workload for a rocprofv3 --kernel-trace pass over the non-headline paths (per-dispatch durations, grouped by kernel and
grid in tools/prof_misc_summary.py): C2 and C4-frame through the fused and the table strategy, BH-7/32 at 2^20, the run-length
kernel (BH-7 2^26 at 16 bits, models cpp and VHDL), C5 parts G = 8 (fused) and G = 4 (both)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import blackman_harris_win_amd as bhw  # noqa: E402
from blackman_harris_win_amd import binding as B  # noqa: E402

out = torch.empty(1 << 26, dtype=torch.int32, device="cuda")
p3 = bhw.make_params(7, 26, 32)
ws = torch.empty(B.lib().bhw_workspace_bytes(ctypes.byref(p3), 0, 1 << 26, B.ALGO_TABLE), dtype=torch.uint8, device="cuda")
for _ in range(200):                       # clock ramp
    bhw.generate(p3, 0, 1 << 26, out=out, workspace=ws)
torch.cuda.synchronize()
N = 30
for win, pw, w in ((4, 20, 24), (4, 16, 24), (7, 20, 32), (7, 16, 32), (5, 18, 24)):
    p = bhw.make_params(win, pw, w)
    for algo in (B.ALGO_FUSED, B.ALGO_TABLE):
        for _ in range(N):
            bhw.generate(p, 0, 1 << pw, out=out, algo=algo, workspace=ws)
        torch.cuda.synchronize()
for model in (B.MODEL_CPP, B.MODEL_VHDL):
    p = bhw.make_params(7, 26, 16, model=model)
    for _ in range(N):
        bhw.generate(p, 0, 1 << 26, out=out, algo=B.ALGO_TABLE, workspace=ws)
    torch.cuda.synchronize()
p = bhw.make_params(4, 24, 14, model=B.MODEL_CPP, combine=B.COMBINE_VHDL)
for _ in range(N):
    bhw.generate(p, 0, 1 << 24, out=out, algo=B.ALGO_TABLE, workspace=ws)
torch.cuda.synchronize()
for G, algo in ((8, B.ALGO_FUSED), (4, B.ALGO_FUSED), (4, B.ALGO_TABLE), (2, B.ALGO_TABLE)):
    for _ in range(N):
        bhw.generate_part(p3, 1, G, out, algo=algo, workspace=ws)
    torch.cuda.synchronize()
print("done")
