#!/usr/bin/env python3
"""Short whole windows inside a HIP graph (20 calls per replay): kernel depth per window without the host's per-call cost.
   python tools/bench_short_graph.py [lib.so ...]      extra libraries (older builds) are timed beside the current one"""
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import blackman_harris_win_amd as bhw  # noqa: E402
from blackman_harris_win_amd import binding as B  # noqa: E402

CASES = [(4, 14, 24), (4, 16, 24), (4, 18, 24), (4, 20, 24), (5, 18, 24), (7, 16, 28), (7, 16, 32), (3, 18, 16), (7, 19, 24), (2, 16, 16), (4, 21, 24), (4, 22, 24), (5, 22, 24), (5, 21, 28), (3, 22, 20)]
if os.environ.get("SHORT_CASES") == "split":                      # where the split form competes with the lockstep / narrow forms
    CASES = [(5, 19, 32), (3, 19, 32), (3, 20, 32), (2, 20, 32), (2, 21, 32), (4, 20, 32), (5, 20, 32), (7, 14, 32), (7, 16, 32), (7, 17, 32), (7, 18, 32), (7, 19, 32), (7, 17, 24), (7, 18, 24), (5, 16, 32), (5, 18, 32), (4, 18, 32), (4, 19, 32), (3, 18, 32)]


def replay_us(gen, p, n, out):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        gen(p, n, out)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(20):
                gen(p, n, out)
        for _ in range(20):
            g.replay()
        st.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(50):
                g.replay()
            e1.record(st)
            st.synchronize()
            ts.append(e0.elapsed_time(e1) / (50 * 20) * 1e3)
    return statistics.median(ts)


def main():
    out = torch.empty(1 << 22, dtype=torch.int32, device="cuda")
    big = torch.empty(1 << 26, dtype=torch.int32, device="cuda")
    p3 = bhw.make_params(7, 26, 32)
    for _ in range(300):                                             # clock ramp
        bhw.generate(p3, 0, 1 << 26, out=big)
    torch.cuda.synchronize()
    gens = [("current", lambda p, n, o: bhw.generate(p, 0, n, out=o[:n]))]
    for path in sys.argv[1:]:
        L = ctypes.CDLL(os.path.abspath(path))
        L.bhw_generate_device.restype = ctypes.c_int

        def gen(p, n, o, L=L):
            rc = L.bhw_generate_device(ctypes.byref(p), ctypes.c_int(0), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream),
                                       ctypes.c_uint64(0), ctypes.c_uint64(n), ctypes.c_void_p(o.data_ptr()))
            assert rc == 0, rc
        gens.append((os.path.basename(path), gen))
    print("%-22s" % "window" + "".join("%26s" % g[0][:25] for g in gens) + "   plan")
    for win, pw, w in CASES:
        p = bhw.make_params(win, pw, w)
        bhw.prepare(p)
        n = 1 << pw
        row = [replay_us(g[1], p, n, out) for g in gens]
        print("%-22s" % ("bh%d 2^%d %d-bit" % (win, pw, w)) + "".join("%23.2f us" % v for v in row) + "   " + B.describe_plan(p, 0, n, B.ALGO_AUTO))


if __name__ == "__main__":
    main()
