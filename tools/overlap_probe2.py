#!/usr/bin/env python3
"""Probe (not part of the product): steady state of a two-window pipeline -- the table build of window i + 1 on one stream while
the tile combine of window i runs on another (separate tables) -- against the two kernels back to back on one stream."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import blackman_harris_win_amd as bhw
from blackman_harris_win_amd import binding as B

L = B.lib()
p = bhw.make_params(7, 26, 32)
N = 1 << 26
out = torch.empty(N, dtype=torch.int32, device="cuda")
ws = [torch.empty((1 << 24) * 8, dtype=torch.uint8, device="cuda") for _ in range(2)]
P = ctypes.byref(p)
L.bhw_dbg_table_build.argtypes = [ctypes.POINTER(B.BhwParams), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
L.bhw_dbg_table_combine.argtypes = [ctypes.POINTER(B.BhwParams), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
s0 = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
sp = lambda s: ctypes.c_void_p(s.cuda_stream)
build = lambda st, w: L.bhw_dbg_table_build(P, 0, sp(st), ctypes.c_void_p(w.data_ptr()))
comb = lambda st, w: L.bhw_dbg_table_combine(P, 0, sp(st), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(out.data_ptr()))
bhw.generate(p, 0, N, out=out)                       # settles the table format
for w in ws:
    assert build(s0, w) == 0
torch.cuda.synchronize()
ref = out.clone()

def timed(fn, iters=300):
    for _ in range(100): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

def serial():
    build(s0, ws[0]); comb(s0, ws[0])

def both():
    build(s1, ws[1]); comb(s2, ws[0])

print("one stream, build + combine     : %.4f ms per window" % timed(serial))
print("build alone                     : %.4f ms" % timed(lambda: build(s1, ws[1])))
print("combine alone                   : %.4f ms" % timed(lambda: comb(s2, ws[0])))
print("two streams, build || combine   : %.4f ms per window" % timed(both))
for pa, pb in ((-1, 0), (0, -1)):
    s1, s2 = torch.cuda.Stream(priority=pa), torch.cuda.Stream(priority=pb)
    print("  priorities build %d combine %d  : %.4f ms per window" % (pa, pb, timed(both)))
# the dependency structure of a real pipeline: build(i+1) may overlap combine(i), but combine(i+1) waits for build(i+1) and
# build(i+2) for combine(i) (two table buffers)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
state = {"i": 0, "built": [None, None], "used": [None, None]}
def piped():
    k = state["i"] & 1; state["i"] += 1
    if state["used"][k] is not None: s1.wait_event(state["used"][k])
    build(s1, ws[k]); eb = torch.cuda.Event(); eb.record(s1)
    s2.wait_event(eb); comb(s2, ws[k]); eu = torch.cuda.Event(); eu.record(s2); state["used"][k] = eu
print("two-buffer pipeline with events : %.4f ms per window" % timed(piped))
assert comb(s0, ws[0]) == 0
torch.cuda.synchronize()
print("parity", bool((out == ref).all()))
