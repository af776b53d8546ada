#!/usr/bin/env python3
"""Timings of the other BASELINE.json configs on one GPU (informative; the contract line is bench.py).
C2: BH-4 N=2^20 24-bit; C3: BH-7 N=2^26 32-bit (both strategies); C4: 1024 x BH-4 N=2^16 24-bit, replicate vs recompute;
C5 shard: BH-7 26/32, one 2^23 shard; sincos sweep 2^26 (model CPP)."""
import json
import time
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import blackman_harris_win_amd as bhw  # noqa: E402
from blackman_harris_win_amd import binding as B  # noqa: E402


def timeit(fn, iters=20, warm=3):
    """Median of five event-timed batches; every batch runs at least `iters` calls and 40 ms (short batches on a cold box read
    several per cent slow: the clocks ramp with the load)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    iters = max(iters, int(40.0 / max(e0.elapsed_time(e1), 1e-3)))
    times = []
    for _ in range(5):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / iters)
    return sorted(times)[2]


def main():
    res = {}
    pr = bhw.make_params(7, 26, 32)
    orr = torch.empty(1 << 26, dtype=torch.int32, device="cuda")
    t0 = time.time()
    while time.time() - t0 < 1.0:                                    # clock ramp, as bench.py
        for _ in range(100):
            bhw.generate(pr, 0, 1 << 26, out=orr)
        torch.cuda.synchronize()
    del orr
    p2 = bhw.make_params(4, 20, 24)
    o2 = torch.empty(1 << 20, dtype=torch.int32, device="cuda")
    for name, algo in (("direct", B.ALGO_DIRECT), ("table", B.ALGO_TABLE)):
        ms = timeit(lambda: bhw.generate(p2, 0, 1 << 20, out=o2, algo=algo))
        res[f"C2_bh4_2^20_24bit_{name}"] = {"ms": ms, "Gsamples/s": (1 << 20) / ms / 1e6}
    p3 = bhw.make_params(7, 26, 32)
    o3 = torch.empty(1 << 26, dtype=torch.int32, device="cuda")
    for name, algo, it in (("direct", B.ALGO_DIRECT, 3), ("table", B.ALGO_TABLE, 20)):
        ms = timeit(lambda: bhw.generate(p3, 0, 1 << 26, out=o3, algo=algo), iters=it, warm=1)
        res[f"C3_bh7_2^26_32bit_{name}"] = {"ms": ms, "Gsamples/s": (1 << 26) / ms / 1e6}
    for name, algo in (("direct", B.ALGO_DIRECT), ("table", B.ALGO_TABLE)):
        ms = timeit(lambda: bhw.generate(p3, 3 << 23, 1 << 23, out=o3, algo=algo), iters=5, warm=1)
        res[f"C5_shard_2^23_of_2^26_{name}"] = {"ms": ms, "Gsamples/s": (1 << 23) / ms / 1e6}
    p4 = bhw.make_params(4, 16, 24)
    o4 = torch.empty((1024, 1 << 16), dtype=torch.int32, device="cuda")
    ms = timeit(lambda: bhw.generate_batched(p4, 1024, out=o4))
    res["C4_1024x_bh4_2^16_replicate"] = {"ms": ms, "Gsamples/s": (1 << 26) / ms / 1e6, "GB/s": 4 * (1 << 26) / ms / 1e6}
    ms = timeit(lambda: bhw.generate(p4, 0, 1 << 26, out=o4.view(-1), algo=B.ALGO_DIRECT), iters=3, warm=1)
    res["C4_1024x_bh4_2^16_recompute_direct"] = {"ms": ms, "Gsamples/s": (1 << 26) / ms / 1e6}
    x = torch.randint(-(1 << 31), (1 << 31) - 1, (1 << 26,), dtype=torch.int32, device="cuda")
    ms = timeit(lambda: bhw.apply(p3, x, out=o3, shift=31), iters=10, warm=2)
    res["fused_apply_bh7_2^26_32bit"] = {"ms": ms, "Gsamples/s": (1 << 26) / ms / 1e6,
                                         "note": "y = (x*w) >> 31, reads x, writes y, no coefficient vector in HBM"}
    for name, w, model in (("cpp_16bit", 16, B.MODEL_CPP), ("vhdl_16bit", 16, B.MODEL_VHDL), ("cpp_24bit", 24, B.MODEL_CPP)):
        pn = bhw.make_params(7, 26, w, model=model)
        ms = timeit(lambda: bhw.generate(pn, 0, 1 << 26, out=o3, algo=B.ALGO_TABLE), iters=10, warm=2)
        res[f"narrow_bh7_2^26_{name}_table"] = {"ms": ms, "Gsamples/s": (1 << 26) / ms / 1e6, "GB/s": 4 * (1 << 26) / ms / 1e6,
                                               "table_entries": 1 << (w - 2)}
    for name, win, w in (("hamming_16bit", 1, 16), ("bh3_24bit", 3, 24), ("hamming_32bit", 1, 32)):
        pt = bhw.make_params(win, 26, w, sin_type=B.SIN_TAYLOR, combine=B.COMBINE_VHDL, lut_size=9)
        ms = timeit(lambda: bhw.generate(pt, 0, 1 << 26, out=o3), iters=10, warm=2)
        res[f"taylor_{name}_2^26"] = {"ms": ms, "Gsamples/s": (1 << 26) / ms / 1e6, "GB/s": 4 * (1 << 26) / ms / 1e6}
    for name, w, comb in (("hlsrule_32bit", 32, B.COMBINE_HLS), ("vhdlrule_32bit", 32, B.COMBINE_VHDL), ("hlsrule_16bit", 16, B.COMBINE_HLS)):
        pt = bhw.make_params(7, 26, w, sin_type=B.SIN_TAYLOR_ALL, combine=comb, lut_size=9)
        ms = timeit(lambda: bhw.generate(pt, 0, 1 << 26, out=o3), iters=10, warm=2)
        res[f"taylor_all_bh7_{name}_2^26"] = {"ms": ms, "Gsamples/s": (1 << 26) / ms / 1e6, "GB/s": 4 * (1 << 26) / ms / 1e6,
                                              "note": "extension BHW_SIN_TAYLOR_ALL (no reference counterpart)"}
    for name, model in (("dds48", B.MODEL_DDS48), ("scaled", B.MODEL_SCALED)):
        pv = bhw.make_params(1, 26, 32, model=model)
        ms = timeit(lambda: bhw.cordic(pv, 0, 1 << 26), iters=3, warm=1)
        res[f"sincos_{name}_2^26_32bit"] = {"ms": ms, "Gphases/s": (1 << 26) / ms / 1e6}
    ax = torch.randint(-(1 << 22), 1 << 22, (1 << 26,), dtype=torch.int32, device="cuda")
    ay = torch.randint(-(1 << 22), 1 << 22, (1 << 26,), dtype=torch.int32, device="cuda")
    ms = timeit(lambda: bhw.atan2(ax, ay, PRECISION=2, INPUT_WIDTH=23, ANGLE_WIDTH=24, out=o3), iters=3, warm=1)
    res["atan2_2^26_p2_23_24"] = {"ms": ms, "Gpairs/s": (1 << 26) / ms / 1e6, "GB/s": 12 * (1 << 26) / ms / 1e6}
    del ax, ay
    pc = bhw.make_params(1, 26, 32, model=B.MODEL_CPP)
    ms = timeit(lambda: bhw.cordic(pc, 0, 1 << 26), iters=5, warm=1)
    res["sincos_cpp_2^26_32bit"] = {"ms": ms, "Gphases/s": (1 << 26) / ms / 1e6}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
