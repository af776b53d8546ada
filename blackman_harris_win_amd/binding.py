"""ctypes binding of include/bhw.h (the C ABI).  Fails loudly when libbhw.so is absent."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))

MODEL_HLS, MODEL_CPP, MODEL_VHDL, MODEL_DDS48, MODEL_SCALED = 0, 1, 2, 3, 4
COMBINE_HLS, COMBINE_VHDL = 0, 1
SIN_CORDIC, SIN_TAYLOR, SIN_TAYLOR_ALL = 0, 1, 2
WIN_HAMMING, WIN_HANN, WIN_BH3, WIN_BH4, WIN_BH5, WIN_BH7 = 1, 2, 3, 4, 5, 7
ALGO_AUTO, ALGO_DIRECT, ALGO_TABLE, ALGO_FUSED = 0, 1, 2, 3
TABLE_BEST, TABLE_PLAIN, TABLE_DELTA16, TABLE_RESIDUAL, TABLE_NIBBLE, TABLE_NIBBLE_ESC = 0, 1, 2, 3, 4, 5
ABI_VERSION = 4

# every symbol include/bhw.h declares (tests check the library exports all of them)
ABI_SYMBOLS = (
    "bhw_abi_version", "bhw_strerror", "bhw_last_error", "bhw_params_init", "bhw_params_validate",
    "bhw_coeffs_from_float", "bhw_constant_tables", "bhw_generate_device", "bhw_generate_device_ex",
    "bhw_workspace_bytes", "bhw_generate_batched_device", "bhw_sincos_device", "bhw_generate_to_host",
    "bhw_sincos_to_host", "bhw_release_device", "bhw_apply_device", "bhw_atan2_device", "bhw_atan2_to_host",
    "bhw_prepare_device", "bhw_part_segments", "bhw_generate_part_device", "bhw_describe_plan",
    "bhw_coeffs_preset", "bhw_gather_parts_device", "bhw_workspace_bytes_ex",
)


class BhwError(RuntimeError):
    def __init__(self, code, detail):
        super().__init__(f"bhw error {code}: {detail}")
        self.code = code
        self.detail = detail


class BhwParams(ctypes.Structure):
    """struct bhw_params of include/bhw.h (the win_selector parameter surface)."""
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("model", ctypes.c_uint32), ("combine", ctypes.c_uint32),
        ("sin_type", ctypes.c_uint32), ("win_type", ctypes.c_uint32), ("n_terms", ctypes.c_uint32),
        ("phi_width", ctypes.c_uint32), ("dat_width", ctypes.c_uint32), ("precision", ctypes.c_uint32),
        ("lut_size", ctypes.c_uint32), ("aa", ctypes.c_int32 * 7),
    ]


class BhwAtan2Params(ctypes.Structure):
    """struct bhw_atan2_params of include/bhw.h (generics of entity cordic_atan2, src/cordic_atan2.vhd:64-69)."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("precision", ctypes.c_uint32),
                ("input_width", ctypes.c_uint32), ("angle_width", ctypes.c_uint32)]


class BhwSegment(ctypes.Structure):
    _fields_ = [("n0", ctypes.c_uint64), ("count", ctypes.c_uint64)]


class BhwExec(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("algo", ctypes.c_uint32),
                ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_uint64),
                ("event_after_build", ctypes.c_void_p), ("table_format", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


_lib = None


def lib_path():
    return os.path.join(HERE, "libbhw.so")


def lib():
    """The loaded C-ABI library.  Raises (no fallback) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -m blackman_harris_win_amd._build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # torch owns the device memory and streams handed to the ABI, so its HIP runtime must be the one this
    # process uses: import it before libbhw.so pulls in libamdhip64 (two runtimes in one process do not
    # share devices or pointers).  A C/C++ host that links libbhw.so directly needs no torch.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(path)
    P = ctypes.POINTER(BhwParams)
    u32, u64, i32p, vp, ci = ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int
    L.bhw_abi_version.restype = u32
    L.bhw_strerror.restype = ctypes.c_char_p
    L.bhw_strerror.argtypes = [ci]
    L.bhw_last_error.restype = ctypes.c_char_p
    L.bhw_params_init.argtypes = [P, u32, u32, u32]
    L.bhw_params_validate.argtypes = [P]
    L.bhw_coeffs_from_float.argtypes = [u32, u32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)]
    L.bhw_constant_tables.argtypes = [u32, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    L.bhw_coeffs_preset.argtypes = [u32, u32, ctypes.POINTER(u32), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)]
    L.bhw_gather_parts_device.argtypes = [P, u32, ctypes.POINTER(ci), ctypes.POINTER(vp), ci, vp, i32p]
    L.bhw_generate_device.argtypes = [P, ci, vp, u64, u64, i32p]
    L.bhw_generate_device_ex.argtypes = [P, ci, vp, u64, u64, i32p, ctypes.POINTER(BhwExec)]
    L.bhw_workspace_bytes.restype = u64
    L.bhw_workspace_bytes.argtypes = [P, u64, u64, u32]
    L.bhw_workspace_bytes_ex.restype = u64
    L.bhw_workspace_bytes_ex.argtypes = [P, u64, u64, ctypes.POINTER(BhwExec)]
    L.bhw_generate_batched_device.argtypes = [P, ci, vp, u32, i32p]
    L.bhw_sincos_device.argtypes = [P, ci, vp, u64, u64, i32p, i32p]
    L.bhw_generate_to_host.argtypes = [P, ci, u64, u64, i32p]
    L.bhw_sincos_to_host.argtypes = [P, ci, u64, u64, i32p, i32p]
    L.bhw_release_device.argtypes = [ci]
    L.bhw_prepare_device.argtypes = [P, ci, vp]
    L.bhw_describe_plan.argtypes = [P, u64, u64, ctypes.POINTER(BhwExec), ctypes.c_char_p, u64]
    L.bhw_part_segments.argtypes = [P, u32, u32, ctypes.POINTER(BhwSegment), u32, ctypes.POINTER(u32)]
    L.bhw_generate_part_device.argtypes = [P, ci, vp, u32, u32, i32p, ctypes.POINTER(BhwExec)]
    L.bhw_apply_device.argtypes = [P, ci, vp, u64, u64, i32p, i32p, u32]
    PA = ctypes.POINTER(BhwAtan2Params)
    L.bhw_atan2_device.argtypes = [PA, ci, vp, u64, i32p, i32p, i32p]
    L.bhw_atan2_to_host.argtypes = [PA, ci, u64, i32p, i32p, i32p]
    if L.bhw_abi_version() != ABI_VERSION:
        raise ImportError(f"{path} has ABI version {L.bhw_abi_version()}, this binding needs {ABI_VERSION}: rebuild it "
                          "(`python -m blackman_harris_win_amd._build --force`)")
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise BhwError(rc, lib().bhw_last_error().decode(errors="replace"))


def describe_plan(params, n0, count, algo=ALGO_AUTO, table_format=TABLE_BEST):
    """One line: strategy, table format and kernel names a call would launch now (bhw_describe_plan)."""
    ex = BhwExec()
    ex.struct_size = ctypes.sizeof(BhwExec)
    ex.algo = algo
    ex.table_format = table_format
    buf = ctypes.create_string_buffer(256)
    check(lib().bhw_describe_plan(ctypes.byref(params), int(n0), int(count), ctypes.byref(ex), buf, 256))
    return buf.value.decode()


def part_segments(params, part, n_parts):
    """[(n0, count), ...]: the coefficients interleaved-ownership part `part` of `n_parts` owns (bhw_part_segments).  Host arithmetic."""
    n = ctypes.c_uint32()
    segs = (BhwSegment * 256)()
    check(lib().bhw_part_segments(ctypes.byref(params), part, n_parts, segs, 256, ctypes.byref(n)))
    return [(int(segs[i].n0), int(segs[i].count)) for i in range(n.value)]


def coeffs_from_float(win_type, dat_width, a=None):
    """a_k = round(coe_k * (2^(W-s) - 1)) -- hls/windows/win_function.cpp:176-177,...,349-355."""
    aa = (ctypes.c_int32 * 7)()
    arr = None
    if a is not None:
        arr = (ctypes.c_double * 7)(*(list(a) + [0.0] * (7 - len(a))))
    check(lib().bhw_coeffs_from_float(win_type, dat_width, arr, aa))
    return list(aa)


PRESETS = {"nuttall": 1, "blackman-nuttall": 2, "flat-top-1": 3, "flat-top-2": 4, "bh7-readme": 5, "blackman": 6, "bh3": 7}


def coeffs_preset(name, dat_width):
    """(win_type, float weights, integer weights) of a named coefficient set the reference lists beside its built-ins
    (bhw_coeffs_preset: hls/windows/win_function.cpp:241-250,292-303, README.md:30-51)."""
    wt = ctypes.c_uint32(0)
    a = (ctypes.c_double * 7)()
    aa = (ctypes.c_int32 * 7)()
    check(lib().bhw_coeffs_preset(PRESETS[name] if isinstance(name, str) else int(name), dat_width, ctypes.byref(wt), a, aa))
    return int(wt.value), list(a), list(aa)


def constant_tables(which):
    t = (ctypes.c_int64 * 48)()
    g = (ctypes.c_int64 * 2)()
    check(lib().bhw_constant_tables(which, t, g))
    return list(t), list(g)


def make_params(win_type, phi_width, dat_width, *, model=MODEL_HLS, combine=COMBINE_HLS, sin_type=SIN_CORDIC,
                precision=1, lut_size=9, aa=None, n_terms=None, validate=True):
    """Build a bhw_params.  `aa` overrides the built-in integer weights (the AA0..AA6 ports)."""
    p = BhwParams()
    rc = lib().bhw_params_init(ctypes.byref(p), win_type, phi_width, dat_width)
    # bhw_params_init validates with the defaults (model HLS); re-validate below with the caller's choices
    if rc != 0 and p.n_terms == 0:
        check(rc)
    p.model, p.combine, p.sin_type = model, combine, sin_type
    p.precision, p.lut_size = precision, lut_size
    if n_terms is not None:
        p.n_terms = n_terms
    if aa is not None:
        vals = list(aa) + [0] * (7 - len(aa))
        for k in range(7):
            p.aa[k] = int(vals[k])
    # the variant generators (cordic_dds48 / cordic_dds_scaled) are sin/cos sources only: bhw_sincos_* validates them
    if validate and model <= MODEL_VHDL:
        check(lib().bhw_params_validate(ctypes.byref(p)))
    return p
