"""ctypes access to the test oracle (oracle/liboracle.so) and to oracle/_ref.

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes
import glob
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_DIR = os.path.join(ROOT, "oracle", "_ref")

MODEL_HLS, MODEL_CPP, MODEL_VHDL, MODEL_DDS48, MODEL_SCALED = 0, 1, 2, 3, 4
COMBINE_HLS, COMBINE_VHDL = 0, 1
SIN_CORDIC, SIN_TAYLOR, SIN_TAYLOR_ALL = 0, 1, 2
TERMS = {1: 2, 2: 2, 3: 3, 4: 4, 5: 5, 7: 7}


class OParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in
                "model combine sin_type n_terms phi_width dat_width precision lut_size".split()] + [("aa", ctypes.c_int32 * 7)]


_o = None


def oracle():
    global _o
    if _o is None:
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError(f"{ORACLE_SO} missing: run `make -C oracle`")
        o = ctypes.CDLL(ORACLE_SO)
        PP = ctypes.POINTER(OParams)
        u32, u64, vp = ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p
        o.bhwo_generate.argtypes = [PP, u64, u64, vp]
        o.bhwo_sincos.argtypes = [PP, u64, u64, vp, vp]
        o.bhwo_cordic.argtypes = [u32, u32, u32, u32, u64, vp, vp, ctypes.POINTER(u64)]
        o.bhwo_taylor.argtypes = [u32, u32, u32, u64, vp, vp]
        o.bhwo_coeffs_from_float.argtypes = [u32, u32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)]
        o.bhwo_fnv1a64.restype = u64
        o.bhwo_fnv1a64.argtypes = [vp, u64, u64]
        o.bhwo_table_t2.restype = ctypes.POINTER(ctypes.c_int64)
        o.bhwo_table_t4.restype = ctypes.POINTER(ctypes.c_int64)
        o.bhwo_gain46.restype = ctypes.c_int64
        o.bhwo_gain47.restype = ctypes.c_int64
        _o = o
    return _o


def oparams(win_type, phi_width, dat_width, *, model=MODEL_HLS, combine=COMBINE_HLS, sin_type=SIN_CORDIC,
            precision=1, lut_size=9, aa=None, n_terms=None):
    p = OParams()
    p.model, p.combine, p.sin_type = model, combine, sin_type
    p.n_terms = n_terms if n_terms is not None else TERMS[win_type]
    p.phi_width, p.dat_width, p.precision, p.lut_size = phi_width, dat_width, precision, lut_size
    if aa is None:
        arr = (ctypes.c_int32 * 7)()
        assert oracle().bhwo_coeffs_from_float(win_type, dat_width, None, arr) == 0
        aa = list(arr)
    vals = list(aa) + [0] * (7 - len(aa))
    for k in range(7):
        p.aa[k] = int(vals[k])
    return p


def from_bhw(bp):
    """OParams carrying the same fields as a product BhwParams (so both sides see one parameter set)."""
    p = OParams()
    for f in "model combine sin_type n_terms phi_width dat_width precision lut_size".split():
        setattr(p, f, getattr(bp, f))
    for k in range(7):
        p.aa[k] = bp.aa[k]
    return p


def generate(p, n0, count):
    out = np.empty(int(count), np.int32)
    rc = oracle().bhwo_generate(ctypes.byref(p), int(n0), int(count), out.ctypes.data)
    if rc:
        raise ValueError("oracle rejected the parameters")
    return out


_cb = None


def generate_mt(p, n0, count, threads=None):
    """bhwo_generate over host threads (oracle/libcpubaseline.so, the same restatement): for whole windows of 2^22 and up."""
    if threads is None:
        threads = host_threads()
    out = np.empty(int(count), np.int32)
    if _cpubaseline().bhw_cpu_baseline(None, ctypes.byref(p), int(n0), int(count), threads, out.ctypes.data) < 0:
        raise ValueError("oracle rejected the parameters")
    return out


def _cpubaseline():
    global _cb
    if _cb is None:
        _cb = ctypes.CDLL(os.path.join(os.path.dirname(ORACLE_SO), "libcpubaseline.so"))
        _cb.bhw_cpu_baseline.restype = ctypes.c_double
        _cb.bhw_cpu_baseline.argtypes = [ctypes.c_char_p, ctypes.POINTER(OParams), ctypes.c_uint64, ctypes.c_uint64,
                                         ctypes.c_int, ctypes.c_void_p]
        _cb.bhw_cpu_sincos_mt.argtypes = [ctypes.POINTER(OParams), ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p]
    return _cb


def host_threads(cap=16):
    return max(1, min(cap, len(os.sched_getaffinity(0))))


def sincos_mt(p, theta0, count, threads=None):
    """bhwo_sincos over host threads: (sin, cos) for whole quadrants of 2^22 phases and up."""
    s = np.empty(int(count), np.int32)
    c = np.empty(int(count), np.int32)
    if _cpubaseline().bhw_cpu_sincos_mt(ctypes.byref(p), int(theta0), int(count), threads or host_threads(),
                                        s.ctypes.data, c.ctypes.data):
        raise ValueError("oracle rejected the parameters")
    return s, c


def reference_window(p, n0, count, threads=None):
    """The window as the reference's own compiled code evaluates it (oracle/_ref): K-1 calls of cordic() of
    cpp/cordic_sincos.cpp -- compiled from the reference source at (PHASE_WIDTH, DATA_WIDTH) = (p.phi_width, p.dat_width),
    oracle/Makefile -- per coefficient, followed by the cosine-sum of hls/windows/win_function.cpp:361-375 (oracle/cpu_baseline.c
    ref_worker).  No restatement of the CORDIC is involved.  Raises FileNotFoundError when that width pair was not built."""
    lib = os.path.join(REF_DIR, f"libref_cordic_{p.phi_width}_{p.dat_width}.so")
    if not os.path.exists(lib):
        raise FileNotFoundError(lib)
    out = np.empty(int(count), np.int32)
    dt = _cpubaseline().bhw_cpu_baseline(lib.encode(), ctypes.byref(p), int(n0), int(count), threads or host_threads(), out.ctypes.data)
    if dt < 0:
        raise RuntimeError(f"bhw_cpu_baseline failed ({dt}) for {lib}")
    return out


def sincos(p, theta0, count):
    s = np.empty(int(count), np.int32)
    c = np.empty(int(count), np.int32)
    rc = oracle().bhwo_sincos(ctypes.byref(p), int(theta0), int(count), s.ctypes.data, c.ctypes.data)
    if rc:
        raise ValueError("oracle rejected the parameters")
    return s, c


def atan2(precision, input_width, angle_width, x, y):
    """cordic_atan2 restatement over integer arrays x, y."""
    o = oracle()
    o.bhwo_atan2.argtypes = [ctypes.c_uint32] * 3 + [ctypes.c_int64] * 2 + [ctypes.c_void_p]
    one = np.empty(1, np.int32)
    out = np.empty(len(x), np.int32)
    for i, (a, b) in enumerate(zip(x, y)):
        if o.bhwo_atan2(precision, input_width, angle_width, int(a), int(b), one.ctypes.data):
            raise ValueError("oracle rejected the atan2 parameters")
        out[i] = one[0]
    return out


def taylor(phi_width, dat_width, lut_size, phases):
    """taylor_sincos generator swept over `phases` (any integer array): returns (cos, sin)."""
    o = oracle()
    c1 = np.empty(1, np.int32)
    s1 = np.empty(1, np.int32)
    c = np.empty(len(phases), np.int32)
    s = np.empty(len(phases), np.int32)
    for i, t in enumerate(phases):
        assert o.bhwo_taylor(phi_width, dat_width, lut_size, int(t), c1.ctypes.data, s1.ctypes.data) == 0
        c[i], s[i] = c1[0], s1[0]
    return c, s


def coeffs(win_type, dat_width, a=None):
    arr = (ctypes.c_int32 * 7)()
    fa = None if a is None else (ctypes.c_double * 7)(*(list(a) + [0.0] * (7 - len(a))))
    assert oracle().bhwo_coeffs_from_float(win_type, dat_width, fa, arr) == 0
    return list(arr)


def fnv(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return "%016x" % oracle().bhwo_fnv1a64(a.ctypes.data, a.size, 0)


def tables():
    o = oracle()
    t2, t4 = o.bhwo_table_t2(), o.bhwo_table_t4()
    return [t2[i] for i in range(48)], [t4[i] for i in range(48)], o.bhwo_gain46(), o.bhwo_gain47()


# ---- oracle/_ref: the reference's own cordic() (cpp/cordic_sincos.cpp:10), built per (PW, W) ----
def ref_pairs():
    out = []
    for f in sorted(glob.glob(os.path.join(REF_DIR, "libref_cordic_*.so"))):
        m = re.search(r"_(\d+)_(\d+)\.so$", f)
        out.append((int(m.group(1)), int(m.group(2)), f))
    return out


class RefCordic:
    """void cordic(int theta, long long *lut, int *s, int *c) of the reference, at fixed widths."""

    def __init__(self, path):
        self.lib = ctypes.CDLL(path)
        self.fn = self.lib._Z6cordiciPxPiS0_
        self.fn.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_int),
                            ctypes.POINTER(ctypes.c_int)]
        t2 = tables()[0]  # the 48-entry ROM main() passes in (cpp/cordic_sincos.cpp:97-110,137)
        self.lut = (ctypes.c_longlong * 48)(*t2)

    def __call__(self, theta):
        s, c = ctypes.c_int(), ctypes.c_int()
        self.fn(int(theta), self.lut, ctypes.byref(s), ctypes.byref(c))
        return s.value, c.value

    def sweep(self, thetas):
        s = np.empty(len(thetas), np.int32)
        c = np.empty(len(thetas), np.int32)
        si, ci = ctypes.c_int(), ctypes.c_int()
        for i, th in enumerate(thetas):
            self.fn(int(th), self.lut, ctypes.byref(si), ctypes.byref(ci))
            s[i], c[i] = si.value, ci.value
        return s, c
