"""Timeline of the combine pass (k_table_combine_tile) from in-kernel wall-clock stamps: when workgroups start and end, how long a
tile lives, how the last round ends -- beside the pass's event-to-event time.  GPU box; needs a library built with
-DBHW_COMBINE_STAMPS (AB_UNITS=bhw_combine.hip python tools/ab_inproc.py --build-only "" "-DBHW_COMBINE_STAMPS").
    python tools/combine_timeline.py "-DBHW_COMBINE_STAMPS"
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from blackman_harris_win_amd import binding  # noqa: E402
import ab_inproc  # noqa: E402


def main():
    os.environ.setdefault("AB_UNITS", "bhw_combine.hip")
    for flags in sys.argv[1:]:
        L = ctypes.CDLL(os.path.join(ROOT, flags[4:]) if flags.startswith("lib:") else ab_inproc.build_variant(0, flags))
        L.bhw_generate_device.argtypes = [ctypes.POINTER(binding.BhwParams), ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
        L.bhw_generate_device_ex.argtypes = [ctypes.POINTER(binding.BhwParams), ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                                             ctypes.c_void_p, ctypes.POINTER(binding.BhwExec)]
        L.bhw_params_init.argtypes = [ctypes.POINTER(binding.BhwParams), ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        L.bhw_dbg_combine_stamps.argtypes = [ctypes.c_void_p]
        p = binding.BhwParams()
        L.bhw_params_init(ctypes.byref(p), 7, 26, 32)
        p.model = int(os.environ.get("AB_MODEL", "0"))
        n = 1 << 26
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(300):
            assert L.bhw_generate_device(ctypes.byref(p), 0, ctypes.c_void_p(st), 0, n, ctypes.c_void_p(out.data_ptr())) == 0
        torch.cuda.synchronize()
        stamps = torch.zeros(16384 * 8, dtype=torch.int64, device="cuda").view(-1, 8)
        stamps[:, 0] = (1 << 62)
        stamps[:, 2] = (1 << 62)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        for x in e:
            x.record()
        torch.cuda.synchronize()
        assert L.bhw_dbg_combine_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
        ex = binding.BhwExec()
        ex.struct_size = ctypes.sizeof(binding.BhwExec)
        ex.event_after_build = e[1].cuda_event
        e[0].record()
        L.bhw_generate_device_ex(ctypes.byref(p), 0, ctypes.c_void_p(st), 0, n, ctypes.c_void_p(out.data_ptr()), ctypes.byref(ex))
        e[2].record()
        torch.cuda.synchronize()
        L.bhw_dbg_combine_stamps(None)
        t = stamps.cpu().numpy()
        t = t[t[:, 1] != 0].astype(np.float64) * 0.01               # 100 MHz -> us
        t0 = t[:, 0].min()

        def q(v):
            return "min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f" % (v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max())
        print("[%s] %d workgroups; events: build %.1f us, combine %.1f us" % (flags, len(t), e[0].elapsed_time(e[1]) * 1e3, e[1].elapsed_time(e[2]) * 1e3))
        start, end = t[:, 0] - t0, t[:, 4] - t0
        print("  first wave starts (us after the first workgroup) ", q(start))
        print("  launch of a workgroup: last wave - first wave     ", q(t[:, 1] - t[:, 0]))
        print("  life of a workgroup (first start .. last store)   ", q(t[:, 4] - t[:, 0]))
        print("  ... until the first wave reaches its stores       ", q(t[:, 2] - t[:, 0]))
        print("  ... until the last wave reaches its stores        ", q(t[:, 3] - t[:, 0]))
        print("  end of a workgroup                                ", q(end))
        order = np.argsort(start)
        print("  workgroups started in the first 1 / 2 / 5 us: %d / %d / %d" % ((start < 1).sum(), (start < 2).sum(), (start < 5).sum()))
        last = end.max()
        for back in (1, 2, 4, 8, 12):
            print("  workgroups still running %2d us before the last one ends: %d" % (back, ((t[:, 0] - t0 < last - back) & (end > last - back)).sum()))
        life = t[:, 4] - t[:, 0]
        k = len(t) // 6
        print("  life by start order, sixths: " + "  ".join("%.1f" % np.median(life[order[i * k:(i + 1) * k]]) for i in range(6)))
        hw = (t[:, 6] / 0.01).astype(np.int64)
        xcc = (t[:, 7] / 0.01).astype(np.int64) & 15
        cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
        ids, cnt = np.unique(cu, return_counts=True)
        print("  distinct CUs %d, tiles per CU: %s" % (len(ids), dict(zip(*[x.tolist() for x in np.unique(cnt, return_counts=True)]))))
        print("  last store of the pass %.1f us after its first workgroup started" % last)


if __name__ == "__main__":
    main()
