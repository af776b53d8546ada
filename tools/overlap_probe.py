#!/usr/bin/env python3
"""Probe (not part of the product): can the VALU-bound table build and the memory-bound tile combine overlap?
(a) two independent windows on two plain streams; (b) build-only and combine-only loops on two streams with
disjoint CU masks (hipExtStreamCreateWithCUMask)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import blackman_harris_win_amd as bhw
from blackman_harris_win_amd import binding as B

L = B.lib()
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
p = bhw.make_params(7, 26, 32)
N = 1 << 26
out = torch.empty(N, dtype=torch.int32, device="cuda")
ws = [torch.empty((1 << 24) * 8, dtype=torch.uint8, device="cuda") for _ in range(2)]
P = ctypes.byref(p)

def mask_stream(cu_lo, cu_hi, total=256):
    words = (ctypes.c_uint32 * (total // 32))()
    for cu in range(cu_lo, cu_hi):
        words[cu // 32] |= 1 << (cu % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), total // 32, words)
    assert rc == 0, rc
    return st

def sync():
    torch.cuda.synchronize()
    hip.hipDeviceSynchronize()

def loop(fn_pairs, iters):
    sync(); t0 = time.perf_counter()
    for i in range(iters):
        for fn in fn_pairs: fn(i)
    sync()
    return (time.perf_counter() - t0) / iters * 1e3

s0 = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
build = lambda st, w: L.bhw_dbg_table_build(P, 0, st, ctypes.c_void_p(w.data_ptr()))
comb = lambda st, w: L.bhw_dbg_table_combine(P, 0, st, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(out.data_ptr()))
L.bhw_dbg_table_build.argtypes = [ctypes.POINTER(B.BhwParams), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
L.bhw_dbg_table_combine.argtypes = [ctypes.POINTER(B.BhwParams), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
build(s0, ws[0]); build(s0, ws[1]); sync()
print("serial  build+combine     : %.4f ms" % loop([lambda i: build(s0, ws[0]), lambda i: comb(s0, ws[0])], 40))
print("build only                : %.4f ms" % loop([lambda i: build(s0, ws[0])], 40))
print("combine only              : %.4f ms" % loop([lambda i: comb(s0, ws[0])], 40))
for x in (64, 96, 128):
    sa, sb = mask_stream(0, x), mask_stream(x, 256)
    # steady state of a pipeline: build (window i+1) on x CUs while combine (window i) runs on the rest
    t = loop([lambda i: build(sa, ws[1]), lambda i: comb(sb, ws[0])], 40)
    tb = loop([lambda i: build(sa, ws[1])], 40)
    tc = loop([lambda i: comb(sb, ws[0])], 40)
    print("CU mask %3d | %3d : both %.4f ms   build alone %.4f   combine alone %.4f" % (x, 256 - x, t, tb, tc))
