"""Timeline of the table-build pass (k_table_build_mirror) from in-kernel wall-clock stamps: when workgroups start, how long their
serial start, their groups and their deferred chains take.  GPU box; needs a library built with -DBHW_BUILD_STAMPS:
    AB_UNITS=bhw_build.hip python tools/ab_inproc.py --build-only "" "-DBHW_BUILD_STAMPS"      (CPU container)
    python tools/build_timeline.py "-DBHW_BUILD_STAMPS" ["-DBHW_BUILD_STAMPS -DBHW_MIRROR_THREADS=256"]   (GPU box)
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
    os.environ.setdefault("AB_UNITS", "bhw_build.hip")
    for flags in sys.argv[1:]:
        L = ctypes.CDLL(os.path.join(ROOT, flags[4:]) if flags.startswith("lib:") else ab_inproc.build_variant(0, flags))   # "lib:build/ab/x.so": a prebuilt stamped library
        L.bhw_generate_device.argtypes = [ctypes.POINTER(binding.BhwParams), ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
        L.bhw_params_init.argtypes = [ctypes.POINTER(binding.BhwParams), ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        L.bhw_dbg_build_stamps.argtypes = [ctypes.c_void_p]
        p = binding.BhwParams()
        L.bhw_params_init(ctypes.byref(p), 7, 26, 32)
        n = 1 << 26
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(300):                                          # format verdicts + clock ramp
            assert L.bhw_generate_device(ctypes.byref(p), 0, ctypes.c_void_p(st), 0, n, ctypes.c_void_p(out.data_ptr())) == 0
        torch.cuda.synchronize()
        stamps = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
        assert L.bhw_dbg_build_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
        L.bhw_generate_device(ctypes.byref(p), 0, ctypes.c_void_p(st), 0, n, ctypes.c_void_p(out.data_ptr()))
        torch.cuda.synchronize()
        L.bhw_dbg_build_stamps(None)
        t = stamps.cpu().numpy().reshape(-1, 16)
        t = t[t[:, 0] != 0].astype(np.float64) * 0.01               # 100 MHz -> us
        t0 = t[:, 0].min()
        rel = t - t0

        def q(v):
            return "min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f" % (v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max())
        print("[%s] %d workgroups, us relative to the first workgroup's start" % (flags, len(t)))
        print("  start of a workgroup              ", q(rel[:, 0]))
        print("  serial start (to the 1st barrier) ", q(t[:, 1] - t[:, 0]))
        print("    ... kernel arguments arrived     ", q(t[:, 8] - t[:, 0]))
        print("    ... tail tables filled           ", q(t[:, 9] - t[:, 0]))
        print("    ... head chains done             ", q(t[:, 10] - t[:, 0]))
        print("    ... group prefixes done          ", q(t[:, 11] - t[:, 0]))
        print("  records (to the 2nd barrier)      ", q(t[:, 2] - t[:, 1]))
        print("  groups, first wave done           ", q(t[:, 3] - t[:, 2]))
        print("  groups, last wave done            ", q(t[:, 4] - t[:, 2]))
        print("  deferred image chains             ", q(t[:, 5] - t[:, 4]))
        print("  whole workgroup                   ", q(t[:, 5] - t[:, 0]))
        print("  end of a workgroup                ", q(rel[:, 5]))
        # where the workgroups ran: HW_ID bits 8..11 CU, 12 SH, 13..15 SE (gfx9), XCC_ID bits 0..3
        hw = t[:, 6] / 0.01
        hw = hw.astype(np.int64)
        xcc = (t[:, 7] / 0.01).astype(np.int64) & 15
        cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
        dur = t[:, 4] - t[:, 2]
        ids, cnt = np.unique(cu, return_counts=True)
        print("  distinct CUs %d, workgroups per CU: %s" % (len(ids), dict(zip(*np.unique(cnt, return_counts=True)))))
        for k in sorted(set(cnt)):
            sel = np.isin(cu, ids[cnt == k])
            print("    CUs with %d workgroup(s): groups phase %s" % (k, q(dur[sel])))
        for x in sorted(set(xcc)):
            print("    XCD %d: %3d workgroups on %2d CUs, groups phase %s" % (x, (xcc == x).sum(), len(set(cu[xcc == x])), q(dur[xcc == x])))
        bi = np.arange(len(t))
        print("    by block index: first 8 %s ... last 8 %s" % (np.round(dur[:8], 1), np.round(dur[-8:], 1)))
        print("    correlation(groups phase, block index) %.2f" % np.corrcoef(bi, dur)[0, 1])


if __name__ == "__main__":
    main()
