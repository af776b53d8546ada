"""In-process A/B of compile-time variants of libbhw.so (GPU box): every variant is built to its own .so, all are
loaded side by side and timed interleaved on the same device and clocks, which removes the box-to-box and
clock-ramp noise of separate bench runs.  usage: python tools/ab_inproc.py "<flags A>" "<flags B>" ...
env: AB_WIN (7) AB_PW (26) AB_W (32) AB_SIN (0) AB_COMBINE (0) AB_MODEL (0) AB_L (9) AB_ROUNDS (8) AB_INNER (100) AB_APPLY AB_PARTS AB_SPLIT
"""
import ctypes
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blackman_harris_win_amd import _build, binding  # noqa: E402


def build_variant(idx, flags):
    # variants are cached under build/ab/ by (flags, source hash): prebuild them in the CPU container with --build-only,
    # the snapshot carries them to the GPU box (GPU minutes are not spent compiling)
    import hashlib
    tag = hashlib.sha256((flags + os.environ.get("AB_UNITS", "") * bool(flags) + _build._digest(_build.library_sources())).encode()).hexdigest()[:16]
    out = os.path.join(ROOT, "build", "ab", f"libbhw_{tag}.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if os.path.exists(out):
        return out
    objdir = os.path.join(ROOT, "build", "ab", "obj_" + tag)
    units = os.environ.get("AB_UNITS")            # e.g. "bhw_build.hip": only these units see the flags, the rest are the "" variant's objects
    if units and flags:
        base = build_variant(idx, "")
        base_tag = os.path.basename(base)[len("libbhw_"):-3]
        objects, _ = _build.compile_units(objdir, flags.split(), only=units.split(","))
        have = {os.path.basename(o) for o in objects}
        base_dir = os.path.join(ROOT, "build", "ab", "obj_" + base_tag)
        objects += [os.path.join(base_dir, f) for f in sorted(os.listdir(base_dir)) if f.endswith(".o") and f not in have]
    else:
        objects, _ = _build.compile_units(objdir, flags.split())
    _build.link_library(objects, out)
    return out


def main():
    variants = [a for a in sys.argv[1:] if a != "--build-only"]
    if "--build-only" in sys.argv:
        for i, f in enumerate(variants):
            print(build_variant(i, f))
        return
    pw = int(os.environ.get("AB_PW", "26"))
    win = int(os.environ.get("AB_WIN", "7"))
    width = int(os.environ.get("AB_W", "32"))
    rounds = int(os.environ.get("AB_ROUNDS", "8"))
    inner = int(os.environ.get("AB_INNER", "100"))
    import torch
    torch.zeros(1, device="cuda")
    libs = []
    # AB_EXTRA_LIBS: comma-separated prebuilt libraries (e.g. an older commit built into build/ab/) timed beside the variants
    extra = [e for e in os.environ.get("AB_EXTRA_LIBS", "").split(",") if e]
    paths = [build_variant(i, f) for i, f in enumerate(variants)] + [os.path.join(ROOT, e) for e in extra]
    variants = variants + ["lib:" + os.path.basename(e) for e in extra]
    for path in paths:
        L = ctypes.CDLL(path)
        L.bhw_generate_device.argtypes = [ctypes.POINTER(binding.BhwParams), ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
        L.bhw_params_init.argtypes = [ctypes.POINTER(binding.BhwParams), ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        if hasattr(L, "bhw_generate_part_device"):
            L.bhw_generate_part_device.argtypes = [ctypes.POINTER(binding.BhwParams), ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32,
                                                   ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        libs.append(L)
    n = 1 << pw
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    p = binding.BhwParams()
    libs[0].bhw_params_init(ctypes.byref(p), win, pw, width)
    p.sin_type = int(os.environ.get("AB_SIN", "0"))
    p.combine = int(os.environ.get("AB_COMBINE", "0"))
    p.model = int(os.environ.get("AB_MODEL", "0"))
    p.lut_size = int(os.environ.get("AB_L", "9"))
    parts = int(os.environ.get("AB_PARTS", "0"))          # AB_PARTS=G: time interleaved ownership part 1 of G instead of the whole window

    apply = os.environ.get("AB_APPLY")                    # AB_APPLY=1: time the fused apply y = (x * w) >> 31 instead of the plain window
    x = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda") if apply else None
    for L in libs if apply else []:
        L.bhw_apply_device.argtypes = [ctypes.POINTER(binding.BhwParams), ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]

    batched = int(os.environ.get("AB_BATCHED", "0"))      # AB_BATCHED=F: F identical frames (bhw_generate_batched_device) into a buffer of F * 2^PW
    if batched:
        out = torch.empty(n * batched, dtype=torch.int32, device="cuda")
        for L in libs:
            L.bhw_generate_batched_device.argtypes = [ctypes.POINTER(binding.BhwParams), ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]

    def call(L):
        if batched:
            return L.bhw_generate_batched_device(ctypes.byref(p), 0, ctypes.c_void_p(st), batched, ctypes.c_void_p(out.data_ptr()))
        if apply:
            return L.bhw_apply_device(ctypes.byref(p), 0, ctypes.c_void_p(st), 0, n, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), 31)
        if parts:
            return L.bhw_generate_part_device(ctypes.byref(p), 0, ctypes.c_void_p(st), 1, parts, ctypes.c_void_p(out.data_ptr()), None)
        return L.bhw_generate_device(ctypes.byref(p), 0, ctypes.c_void_p(st), 0, n, ctypes.c_void_p(out.data_ptr()))

    if os.environ.get("AB_FORCE_NIBBLE"):                 # timing experiments that corrupt the table values: pin the format verdict
        for L in libs:
            d = ctypes.c_uint32(0)
            L.bhw_dbg_table_format_info(ctypes.byref(p), ctypes.byref(d), None)
            L.bhw_dbg_table_format_verdict(ctypes.byref(p), ctypes.c_uint32(16 + d.value), 1)
    ref = None
    for L in libs:
        out.zero_()
        rc = call(L)
        assert rc == 0, rc
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        else:
            assert os.environ.get("AB_NOCHECK") or torch.equal(ref, out), "variants disagree"
    times = [[] for _ in libs]
    for r in range(rounds + 1):
        for i, L in enumerate(libs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(inner):
                call(L)
            e1.record()
            torch.cuda.synchronize()
            if r:   # round 0 = clock ramp
                times[i].append(e0.elapsed_time(e1) / inner)
    for f, t in zip(variants, times):
        print("%-44s median %.4f ms  min %.4f  max %.4f" % ("[" + f + "]", statistics.median(t), min(t), max(t)))
    if os.environ.get("AB_SPLIT"):
        # the two passes of the table strategy on their own: an event between build and combine (bhw_exec.event_after_build),
        # one call at a time (the events of back-to-back calls would serialise the host), median over AB_SPLIT_N calls
        n_split = int(os.environ.get("AB_SPLIT_N", "300"))
        for f, L in zip(variants, libs):
            L.bhw_generate_device_ex.argtypes = [ctypes.POINTER(binding.BhwParams), ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                                                 ctypes.c_void_p, ctypes.POINTER(binding.BhwExec)]
            evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n_split)]
            for trio in evs:
                for e in trio:
                    e.record()
            torch.cuda.synchronize()
            ex = binding.BhwExec()
            ex.struct_size = ctypes.sizeof(binding.BhwExec)
            for _ in range(100):
                call(L)
            for e0, em, e1 in evs:
                ex.event_after_build = em.cuda_event
                e0.record()
                rc = L.bhw_generate_device_ex(ctypes.byref(p), 0, ctypes.c_void_p(st), 0, n, ctypes.c_void_p(out.data_ptr()), ctypes.byref(ex))
                assert rc == 0, rc
                e1.record()
            torch.cuda.synchronize()
            b = [t[0].elapsed_time(t[1]) * 1e3 for t in evs]
            c = [t[1].elapsed_time(t[2]) * 1e3 for t in evs]
            print("%-44s build %.1f us (p10 %.1f p90 %.1f)  combine %.1f us (p10 %.1f p90 %.1f)" % (
                "[" + f + "]", statistics.median(b), sorted(b)[n_split // 10], sorted(b)[9 * n_split // 10],
                statistics.median(c), sorted(c)[n_split // 10], sorted(c)[9 * n_split // 10]))


if __name__ == "__main__":
    main()
