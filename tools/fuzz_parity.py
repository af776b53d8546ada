#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity campaign (run on the GPU box; a fixed-seed slice of it is tests/test_gpu_fuzz.py).
Every case draws a parameter set (all bit-models, both cosine-sum rules, CORDIC and Taylor sources incl. the all-term-count
extension, random integer weights, widths 8..32, lengths 2^4..2^24), a stream range and a strategy, generates it through
the C ABI and compares bit-for-bit with the oracle (threaded via oracle/libcpubaseline.so).  One case in eight exercises
the variant generators instead (cordic_dds48 / cordic_dds_scaled through bhw_sincos_device, cordic_atan2).
usage: fuzz_parity.py <seconds> [seed [max_pw]]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_lib as O
import blackman_harris_win_amd as bhw
from blackman_harris_win_amd import binding as B

CB = ctypes.CDLL(os.path.join(ROOT, "oracle", "libcpubaseline.so"))
CB.bhw_cpu_baseline.restype = ctypes.c_double
CB.bhw_cpu_baseline.argtypes = [ctypes.c_char_p, ctypes.POINTER(O.OParams), ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p]
threads = min(16, len(os.sched_getaffinity(0)))
CLASSES = ["hls/hlsrule", "hls/vhdlrule", "cpp/hlsrule", "cpp/vhdlrule", "vhdl/hlsrule", "vhdl/vhdlrule",
           "taylor/hlsrule", "taylor/vhdlrule", "taylor_all/hlsrule", "taylor_all/vhdlrule", "dds48", "scaled", "atan2"]


class Mismatch(AssertionError):
    pass


def oracle_gen(po, n0, count):
    out = np.empty(count, np.int32)
    assert CB.bhw_cpu_baseline(None, ctypes.byref(po), n0, count, threads, out.ctypes.data) >= 0
    return out


def fuzz(budget=60.0, seed=1, max_cases=None, max_count=1 << 25, max_pw=25, verbose=True):
    """Runs until `budget` seconds or `max_cases` cases (whichever comes first; max_cases alone makes the run deterministic).
    Returns (cases, coefficients, cases by class); raises Mismatch with the failing parameter set."""
    rng = np.random.default_rng(seed)
    t0 = time.time(); cases = 0; samples = 0; by = {}; last = t0
    while (budget is None or time.time() - t0 < budget) and (max_cases is None or cases < max_cases):
        if verbose and time.time() - last > 60:                             # progress line (long silent runs are taken to be hung)
            last = time.time(); print("... %d cases, %d coefficients so far" % (cases, samples), flush=True)
        kind = rng.random()
        if kind < 0.09:                                                     # cordic_dds48 / cordic_dds_scaled
            model = int(rng.choice([B.MODEL_DDS48, B.MODEL_SCALED])); pw = int(rng.integers(4, 31)); w = int(rng.integers(8, 33))
            p = B.make_params(1, pw, w, model=model)
            th0 = int(rng.integers(0, 1 << pw)); cnt = int(rng.integers(1, 20000))
            gs, gc = bhw.cordic(p, th0, cnt)
            ws, wc = O.sincos(O.from_bhw(p), th0, cnt)
            if not (np.array_equal(gs.cpu().numpy(), ws) and np.array_equal(gc.cpu().numpy(), wc)):
                raise Mismatch(dict(model=model, pw=pw, w=w, theta0=th0, count=cnt))
            cases += 1; samples += cnt; key = ("dds48", "scaled")[model - 3]; by[key] = by.get(key, 0) + 1
            continue
        if kind < 0.125:                                                    # cordic_atan2
            AW = int(rng.integers(4, 33)); IW = int(rng.integers(AW - 1, 33)); P = int(rng.integers(1, 8)); cnt = int(rng.integers(1, 4000))
            lo, hi = -(1 << (IW - 1)), (1 << (IW - 1)) - 1
            x = rng.integers(lo, hi + 1, cnt); y = rng.integers(lo, hi + 1, cnt)
            if rng.random() < 0.3: x = x >> int(rng.integers(0, IW)); y = y >> int(rng.integers(0, IW))   # small magnitudes too
            got = bhw.atan2(torch.tensor(x.astype(np.int32), device="cuda"), torch.tensor(y.astype(np.int32), device="cuda"),
                            PRECISION=P, INPUT_WIDTH=IW, ANGLE_WIDTH=AW).cpu().numpy()
            if not np.array_equal(got, O.atan2(P, IW, AW, x, y)):
                raise Mismatch(dict(atan2=(P, IW, AW)))
            cases += 1; samples += cnt; by["atan2"] = by.get("atan2", 0) + 1
            continue
        win = int(rng.choice([1, 2, 3, 4, 5, 7])); K = O.TERMS[win]
        taylor = rng.random() < 0.25
        model = int(rng.integers(0, 3)); combine = int(rng.integers(0, 2))
        w = int(rng.integers(8, 33)); pw = int(rng.integers(4, max_pw + 1))
        if rng.random() < 0.15 and max_pw >= 22: pw = int(rng.integers(22, max_pw + 1))   # whole-period tile calls: packed table formats
        prec = int(rng.integers(1, 4)) if model == B.MODEL_VHDL else 1
        lut = 9
        if taylor:
            pw = int(rng.integers(6, 21)); lut = int(rng.integers(max(1, pw - 17), min(14, pw + 2)))
        elif model == B.MODEL_HLS and pw > w + 2:
            pw = w + 2
        aa = None
        if rng.random() < 0.6:
            aa = [int(v) for v in rng.integers(-(1 << (w - 1)), 1 << (w - 1), 7)]
        try:
            p = B.make_params(win, pw, w, model=model, combine=combine, precision=prec, aa=aa,
                              sin_type=(B.SIN_TAYLOR if (K <= 3 and rng.random() < 0.5) else B.SIN_TAYLOR_ALL) if taylor else B.SIN_CORDIC,
                              lut_size=lut)
        except B.BhwError:
            continue
        n = 1 << pw
        mode = rng.random()
        if mode < 0.35:   n0, count = 0, n                                   # whole period
        elif mode < 0.5:  n0, count = n * int(rng.integers(0, 3)), n * int(rng.integers(1, 3))
        else:             n0, count = int(rng.integers(0, 4 * n)), int(rng.integers(1, min(4 * n, 300000) + 1))
        count = min(count, (1 << 17) if taylor else max_count)              # the Taylor oracle evaluates its ROM in binary128 per sample
        if taylor and count < n <= (1 << 17) and mode < 0.5: n0, count = 0, n
        algo = int(rng.choice([B.ALGO_AUTO, B.ALGO_DIRECT, B.ALGO_TABLE, B.ALGO_FUSED]))
        desc = dict(win=win, pw=pw, w=w, model=model, combine=combine, prec=prec, aa=aa, taylor=taylor, lut=lut, n0=n0, count=count, algo=algo)
        try:
            got = bhw.generate(p, n0, count, algo=algo).cpu().numpy()
        except B.BhwError:
            print("ERROR", desc, flush=True)
            raise
        want = oracle_gen(O.from_bhw(p), n0, count)
        if not np.array_equal(got, want):
            bad = int(np.flatnonzero(got != want)[0])
            raise Mismatch(dict(desc, first_bad=bad, got=int(got[bad]), want=int(want[bad])))
        cases += 1; samples += count
        key = (("taylor" if K <= 3 else "taylor_all") if taylor else ("hls", "cpp", "vhdl")[model]) + "/" + ("hlsrule", "vhdlrule")[combine]
        by[key] = by.get(key, 0) + 1
    return cases, samples, by


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    max_pw = int(sys.argv[3]) if len(sys.argv) > 3 else 25          # e.g. 20: short windows only (the fused kernels' domain)
    t0 = time.time()
    try:
        cases, samples, by = fuzz(budget, seed, max_pw=max_pw)
    except Mismatch as e:
        print("MISMATCH", e.args[0]); sys.exit(1)
    print("fuzz ok: %d cases, %d coefficients compared bit-for-bit in %.0f s (seed %d) on %s" % (cases, samples, time.time() - t0, seed, torch.cuda.get_device_name(0)))
    print("cases by model/rule:", dict(sorted(by.items())))
