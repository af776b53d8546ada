#!/usr/bin/env python3
"""Probe: how much do the VALU-bound table build and the memory-leaning tile combine overlap when two independent
windows are generated on two streams?  (Sizes the build/combine pipelining idea; not part of the product.)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import blackman_harris_win_amd as bhw
from blackman_harris_win_amd import binding as B

p = bhw.make_params(7, 26, 32)
N = 1 << 26
outs = [torch.empty(N, dtype=torch.int32, device="cuda") for _ in range(2)]
wss = [torch.empty((1 << 24) * 8, dtype=torch.uint8, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]

def run(nstreams, iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        for s in range(nstreams):
            with torch.cuda.stream(streams[s]):
                bhw.generate(p, 0, N, out=outs[s], algo=B.ALGO_TABLE, workspace=wss[s])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (iters * nstreams) * 1e3

for _ in range(2):
    print("1 stream : %.4f ms per window" % run(1, 40))
    print("2 streams: %.4f ms per window" % run(2, 40))
