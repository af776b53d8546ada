#!/usr/bin/env python3
"""Round-2 measurements on one GPU (informative; the contract line is bench.py):
(a) short whole windows, 2^9 .. 2^22: fused kernel (one launch) vs table strategy (two launches) vs direct, BH-4/24 and BH-7/32,
    back-to-back calls timed with HIP events (what a caller sees per call) -- the data behind BHW_FUSED_MAX_PW;
(b) C4: first frame + replicate;
(c) C5: one 2^26 BH-7/32 window as G interleaved ownership parts and as G contiguous shards, per-part device time for both
    strategies -- strong-scaling projection (parts are independent: a G-GPU run takes the slowest part's time)."""
import ctypes
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import blackman_harris_win_amd as bhw  # noqa: E402
from blackman_harris_win_amd import binding as B  # noqa: E402


def timeit(fn, iters=200, warm=20, rounds=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    return statistics.median(ts)


def main():
    res = {"small_windows_us": {}, "c4": {}, "c5_parts_ms": {}}
    # clock ramp
    p3 = bhw.make_params(7, 26, 32)
    o3 = torch.empty(1 << 26, dtype=torch.int32, device="cuda")
    for _ in range(300):
        bhw.generate(p3, 0, 1 << 26, out=o3)
    torch.cuda.synchronize()
    ws_big = torch.empty(B.lib().bhw_workspace_bytes(ctypes.byref(p3), 0, 1 << 26, B.ALGO_TABLE), dtype=torch.uint8, device="cuda")
    for win, w in ((4, 24), (7, 32), (7, 16)):
        for pw in (9, 12, 14, 16, 17, 18, 19, 20, 21, 22):
            if pw > w + 2:
                continue
            p = bhw.make_params(win, pw, w)
            n = 1 << pw
            row = {}
            for name, algo in (("fused", B.ALGO_FUSED), ("table", B.ALGO_TABLE), ("direct", B.ALGO_DIRECT), ("auto", B.ALGO_AUTO)):
                if name == "direct" and pw > 20:
                    continue
                it = 200 if pw <= 20 else 50
                row[name] = 1e3 * timeit(lambda: bhw.generate(p, 0, n, out=o3, algo=algo, workspace=ws_big), iters=it)
            res["small_windows_us"][f"bh{win}_{w}bit_2^{pw}"] = row
    # graph replay of the C2 call (launch overhead removed)
    p2 = bhw.make_params(4, 20, 24)
    bhw.prepare(p2)
    for name, algo in (("fused", B.ALGO_FUSED), ("table", B.ALGO_TABLE)):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            g = torch.cuda.CUDAGraph()
            bhw.generate(p2, 0, 1 << 20, out=o3, algo=algo, workspace=ws_big)
            st.synchronize()
            with torch.cuda.graph(g, stream=st):
                for _ in range(20):
                    bhw.generate(p2, 0, 1 << 20, out=o3, algo=algo, workspace=ws_big)
            res["small_windows_us"][f"C2_graph20_{name}"] = 1e3 * timeit(g.replay, iters=20) / 20
    p4 = bhw.make_params(4, 16, 24)
    o4 = torch.empty((1024, 1 << 16), dtype=torch.int32, device="cuda")
    ms = timeit(lambda: bhw.generate_batched(p4, 1024, out=o4), iters=50)
    res["c4"]["1024x_bh4_2^16_replicate_us"] = 1e3 * ms
    res["c4"]["GB/s"] = 4 * (1 << 26) / ms / 1e6
    res["c4"]["first_frame_us"] = 1e3 * timeit(lambda: bhw.generate(p4, 0, 1 << 16, out=o3))
    del o4
    # C5 (the short kernels above let the clocks drop: round 3's file showed 0.1195 ms for AUTO at one part against 0.1123 for the
    # same table plan timed later -- ramp again, and time the strategies of one G interleaved)
    for _ in range(300):
        bhw.generate(p3, 0, 1 << 26, out=o3)
    torch.cuda.synchronize()
    for G in (1, 2, 4, 8):
        row = {}
        for name, algo in (("auto", B.ALGO_AUTO), ("fused", B.ALGO_FUSED), ("table", B.ALGO_TABLE)):
            if name == "fused" and G < 4:
                continue
            per = [timeit(lambda: bhw.generate_part(p3, g, G, o3, algo=algo, workspace=ws_big), iters=30, warm=30, rounds=3) for g in range(G)]
            row[name] = {"max": max(per), "min": min(per)}
        n0c = [(g << 26) // G for g in range(G)]
        per = [timeit(lambda: bhw.generate(p3, n0c[g], (1 << 26) // G, out=o3, workspace=ws_big), iters=30, warm=30, rounds=3) for g in range(G)]
        row["contiguous_auto"] = {"max": max(per), "min": min(per)}
        res["c5_parts_ms"][f"G={G}"] = row
    res["device"] = torch.cuda.get_device_name(0)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
