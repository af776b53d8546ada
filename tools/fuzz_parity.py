#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity campaign (run on the GPU box; not part of the pytest suite).
Every case draws a parameter set (all bit-models, both cosine-sum rules, CORDIC and Taylor sources, random integer
weights, widths 8..32, lengths 2^4..2^24), a stream range and a strategy, generates it through the C ABI and compares
bit-for-bit with the oracle (threaded via oracle/libcpubaseline.so).  usage: fuzz_parity.py <seconds> [seed]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_lib as O
import blackman_harris_win_amd as bhw
from blackman_harris_win_amd import binding as B

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
CB = ctypes.CDLL(os.path.join(ROOT, "oracle", "libcpubaseline.so"))
CB.bhw_cpu_baseline.restype = ctypes.c_double
CB.bhw_cpu_baseline.argtypes = [ctypes.c_char_p, ctypes.POINTER(O.OParams), ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p]
threads = min(16, len(os.sched_getaffinity(0)))

def oracle_gen(po, n0, count):
    out = np.empty(count, np.int32)
    assert CB.bhw_cpu_baseline(None, ctypes.byref(po), n0, count, threads, out.ctypes.data) >= 0
    return out

t0 = time.time(); cases = 0; samples = 0; by = {}
while time.time() - t0 < budget:
    win = int(rng.choice([1, 2, 3, 4, 5, 7])); K = O.TERMS[win]
    taylor = K <= 3 and rng.random() < 0.2
    model = int(rng.integers(0, 3)); combine = int(rng.integers(0, 2))
    w = int(rng.integers(8, 33)); pw = int(rng.integers(4, 25))
    prec = int(rng.integers(1, 4)) if model == B.MODEL_VHDL else 1
    lut = 9
    if taylor:
        pw = int(rng.integers(6, 21)); lut = int(rng.integers(max(1, pw - 17), min(12, pw + 2)))
    elif model == B.MODEL_HLS and pw > w + 2:
        pw = w + 2
    aa = None
    if rng.random() < 0.6:
        aa = [int(v) for v in rng.integers(-(1 << (w - 1)), 1 << (w - 1), 7)]
    try:
        p = B.make_params(win, pw, w, model=model, combine=combine, precision=prec, aa=aa,
                          sin_type=B.SIN_TAYLOR if taylor else B.SIN_CORDIC, lut_size=lut)
    except B.BhwError:
        continue
    n = 1 << pw
    mode = rng.random()
    if mode < 0.35:   n0, count = 0, n                                   # whole period
    elif mode < 0.5:  n0, count = n * int(rng.integers(0, 3)), n * int(rng.integers(1, 3))
    else:             n0, count = int(rng.integers(0, 4 * n)), int(rng.integers(1, min(4 * n, 300000) + 1))
    count = min(count, 1 << 22)
    algo = int(rng.choice([B.ALGO_AUTO, B.ALGO_DIRECT, B.ALGO_TABLE]))
    got = bhw.generate(p, n0, count, algo=algo).cpu().numpy()
    want = oracle_gen(O.from_bhw(p), n0, count)
    if not np.array_equal(got, want):
        bad = int(np.flatnonzero(got != want)[0])
        print("MISMATCH", dict(win=win, pw=pw, w=w, model=model, combine=combine, prec=prec, aa=aa, taylor=taylor, lut=lut,
                               n0=n0, count=count, algo=algo, first_bad=bad, got=int(got[bad]), want=int(want[bad])))
        sys.exit(1)
    cases += 1; samples += count
    key = ("taylor" if taylor else ("hls", "cpp", "vhdl")[model]) + "/" + ("hlsrule", "vhdlrule")[combine]
    by[key] = by.get(key, 0) + 1
print("fuzz ok: %d cases, %d coefficients compared bit-for-bit in %.0f s (seed %d) on %s" % (cases, samples, time.time() - t0, seed, torch.cuda.get_device_name(0)))
print("cases by model/rule:", dict(sorted(by.items())))
