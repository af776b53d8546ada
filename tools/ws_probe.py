#!/usr/bin/env python3
"""Probe: does the table scratch's allocation (torch caching allocator vs hipMalloc inside the library) matter?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import blackman_harris_win_amd as bhw
from blackman_harris_win_amd import binding as B
p = bhw.make_params(7, 26, 32)
N = 1 << 26
out = torch.empty(N, dtype=torch.int32, device="cuda")
ws = torch.empty((1 << 24) * 8, dtype=torch.uint8, device="cuda")
ws2 = torch.empty((1 << 24) * 8 + (1 << 21), dtype=torch.uint8, device="cuda")
off = (-ws2.data_ptr()) % (1 << 21)
ws_aligned = ws2[off:off + (1 << 24) * 8]
print("out ptr %x  ws ptr %x  ws_aligned %x" % (out.data_ptr(), ws.data_ptr(), ws_aligned.data_ptr()))
def t(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for rnd in range(2):
    print("library scratch : %.4f" % t(lambda: bhw.generate(p, 0, N, out=out, algo=B.ALGO_TABLE)))
    print("torch workspace : %.4f" % t(lambda: bhw.generate(p, 0, N, out=out, algo=B.ALGO_TABLE, workspace=ws)))
    print("torch ws 2MB-al : %.4f" % t(lambda: bhw.generate(p, 0, N, out=out, algo=B.ALGO_TABLE, workspace=ws_aligned)))
