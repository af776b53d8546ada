"""Host-side mirror of the reference's operator interface, on top of the C ABI.

torch is plumbing here: it owns the output tensor, the current stream and (in bench.py) the
process group.  The arithmetic is in libbhw.so's HIP kernels.
"""
import ctypes

from . import binding as B

_WIN_TYPES = {
    # WIN_TYPE generic strings of win_selector (src/win_selector.vhd:64,93,115,137,157,178) plus the
    # HLS names (hls/windows/window_test.cpp:59-74)
    "HAMMING": B.WIN_HAMMING, "HANN": B.WIN_HANN,
    "BH3TERM": B.WIN_BH3, "BH4TERM": B.WIN_BH4, "BH5TERM": B.WIN_BH5, "BH7TERM": B.WIN_BH7,
    "Hamming": B.WIN_HAMMING, "Hann": B.WIN_HANN, "Blackman-Harris-3": B.WIN_BH3,
    "Blackman-Harris-4": B.WIN_BH4, "Blackman-Harris-5": B.WIN_BH5, "Blackman-Harris-7": B.WIN_BH7,
}


_TORCH = None


def _torch():
    """torch, once a HIP device has been seen (cached: the per-call cost of the wrappers matters for short windows)."""
    global _TORCH
    if _TORCH is None:
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the window generator runs only on the GPU (no CPU fallback)")
        _TORCH = torch
    return _TORCH


def _stream_ptr(torch, device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _dev_index(torch, device):
    d = torch.device("cuda" if device is None else device)
    return torch.cuda.current_device() if d.index is None else d.index


def shard_range(total, rank, world_size):
    """Contiguous index shard [n0, n0+count) of `total` coefficients for `rank` (no collective)."""
    base, rem = divmod(total, world_size)
    n0 = rank * base + min(rank, rem)
    return n0, base + (1 if rank < rem else 0)


def _check_out(torch, out, need, what="out"):
    """An output tensor handed to the kernels: int32, on a GPU, contiguous, large enough."""
    if out.dtype != torch.int32 or not out.is_cuda or not out.is_contiguous() or out.numel() < need:
        raise ValueError(f"{what} must be a contiguous int32 CUDA tensor with at least {need} elements")
    return out.device.index


def _exec(algo, workspace, event_after_build=None, table_format=B.TABLE_BEST):
    ex = B.BhwExec()
    ex.struct_size = ctypes.sizeof(B.BhwExec)
    ex.algo = algo
    ex.table_format = table_format
    if workspace is not None:
        if not workspace.is_cuda or not workspace.is_contiguous():
            raise ValueError("workspace must be a contiguous CUDA tensor")
        ex.workspace = workspace.data_ptr()
        ex.workspace_bytes = workspace.numel() * workspace.element_size()
    if event_after_build is not None:      # a torch.cuda.Event that has been recorded once (its handle exists)
        ex.event_after_build = event_after_build.cuda_event
    return ex


def prepare(params, *, device=None):
    """Every lazy step of later calls with `params` on the current stream, done now (bhw_prepare_device): Taylor ROM upload,
    library scratch, one-off verification of the packed table formats.  Needed before capturing calls into a HIP graph
    without a caller workspace; otherwise optional."""
    torch = _torch()
    dev = _dev_index(torch, device)
    B.check(B.lib().bhw_prepare_device(ctypes.byref(params), dev, _stream_ptr(torch, dev)))


def generate(params, n0, count, *, device=None, out=None, algo=B.ALGO_AUTO, workspace=None, event_after_build=None,
             table_format=B.TABLE_BEST):
    """count coefficients starting at stream index n0 as an int32 CUDA tensor (bhw_generate_device)."""
    torch = _torch()
    dev = _dev_index(torch, device)
    if out is None:
        out = torch.empty(int(count), dtype=torch.int32, device=f"cuda:{dev}")
    else:
        dev = _check_out(torch, out, int(count))
    if workspace is not None and workspace.device.index != dev:
        raise ValueError("workspace must live on the output's device")
    ex = _exec(algo, workspace, event_after_build, table_format)
    B.check(B.lib().bhw_generate_device_ex(ctypes.byref(params), dev, _stream_ptr(torch, dev), int(n0), int(count),
                                            ctypes.c_void_p(out.data_ptr()), ctypes.byref(ex)))
    return out


def generate_part(params, part, n_parts, window, *, algo=B.ALGO_AUTO, workspace=None, event_after_build=None,
                  table_format=B.TABLE_BEST):
    """Interleaved ownership (bhw_generate_part_device): writes the coefficients part `part` of `n_parts` owns into `window`,
    a full-length (2^phi_width) int32 CUDA tensor; every other element is left untouched.  binding.part_segments lists them."""
    torch = _torch()
    dev = _check_out(torch, window, 1 << params.phi_width, "window")
    if workspace is not None and workspace.device.index != dev:
        raise ValueError("workspace must live on the window's device")
    ex = _exec(algo, workspace, event_after_build, table_format)
    B.check(B.lib().bhw_generate_part_device(ctypes.byref(params), dev, _stream_ptr(torch, dev), int(part), int(n_parts),
                                              ctypes.c_void_p(window.data_ptr()), ctypes.byref(ex)))
    return window


def gather_parts(params, windows, out):
    """One window on out's device from its interleaved ownership parts (bhw_gather_parts_device): windows[g] is the full-length
    int32 CUDA tensor part g of len(windows) was generated into (any device; windows[g] may be `out` itself).  Peer copies of the
    owned segments on out's current stream; the producing streams must have been synchronised with it by the caller."""
    torch = _torch()
    n = 1 << params.phi_width
    dev = _check_out(torch, out, n)
    G = len(windows)
    devs = (ctypes.c_int * G)(*[_check_out(torch, w, n, "window") for w in windows])
    ptrs = (ctypes.c_void_p * G)(*[w.data_ptr() for w in windows])
    B.check(B.lib().bhw_gather_parts_device(ctypes.byref(params), G, devs, ptrs, dev, _stream_ptr(torch, dev),
                                             ctypes.c_void_p(out.data_ptr())))
    return out


def apply(params, x, *, n0=0, shift=None, out=None):
    """Fused apply: y[i] = (x[i] * w[n0+i]) >> shift without materialising w (bhw_apply_device).
    `shift` defaults to dat_width - 1 (unit gain for a full-scale window)."""
    torch = _torch()
    if x.dtype != torch.int32 or not x.is_cuda or not x.is_contiguous():
        raise ValueError("x must be a contiguous int32 CUDA tensor")
    dev = x.device.index
    if out is None:
        out = torch.empty_like(x)
    elif _check_out(torch, out, x.numel()) != dev:
        raise ValueError("out must live on x's device")
    if shift is None:
        shift = params.dat_width - 1
    B.check(B.lib().bhw_apply_device(ctypes.byref(params), dev, _stream_ptr(torch, dev), int(n0), x.numel(),
                                      ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), int(shift)))
    return out


def generate_batched(params, frames, *, device=None, out=None):
    """frames x 2^phi_width coefficients: one period computed, then replicated (bhw_generate_batched_device)."""
    torch = _torch()
    dev = _dev_index(torch, device)
    n = 1 << params.phi_width
    if out is None:
        out = torch.empty((int(frames), n), dtype=torch.int32, device=f"cuda:{dev}")
    dev = _check_out(torch, out, int(frames) * n)
    B.check(B.lib().bhw_generate_batched_device(ctypes.byref(params), dev, _stream_ptr(torch, dev), int(frames),
                                                 ctypes.c_void_p(out.data_ptr())))
    return out


def cordic(params, theta0, count, *, device=None):
    """(sin, cos) int32 CUDA tensors for phases theta0..theta0+count-1 (bhw_sincos_device)."""
    torch = _torch()
    dev = _dev_index(torch, device)
    s = torch.empty(int(count), dtype=torch.int32, device=f"cuda:{dev}")
    c = torch.empty(int(count), dtype=torch.int32, device=f"cuda:{dev}")
    B.check(B.lib().bhw_sincos_device(ctypes.byref(params), dev, _stream_ptr(torch, dev), int(theta0), int(count),
                                       ctypes.c_void_p(s.data_ptr()), ctypes.c_void_p(c.data_ptr())))
    return s, c


def atan2(x, y, *, PRECISION=1, INPUT_WIDTH=20, ANGLE_WIDTH=16, out=None):
    """entity cordic_atan2 (src/cordic_atan2.vhd:64-76) over int32 CUDA tensors VEC_DX = x, VEC_DY = y: returns PHI_DT
    (ANGLE_WIDTH-bit words, sign-extended; full circle = 2^ANGLE_WIDTH) via bhw_atan2_device."""
    torch = _torch()
    if x.dtype != torch.int32 or y.dtype != torch.int32 or not x.is_cuda or x.device != y.device or x.shape != y.shape:
        raise ValueError("x and y must be int32 CUDA tensors of the same shape on one device")
    x, y = x.contiguous(), y.contiguous()
    dev = x.device.index
    phi = torch.empty_like(x) if out is None else out
    if phi.dtype != torch.int32 or phi.device != x.device or phi.numel() != x.numel() or not phi.is_contiguous():
        raise ValueError("out must be a contiguous int32 tensor like x")
    p = B.BhwAtan2Params(ctypes.sizeof(B.BhwAtan2Params), PRECISION, INPUT_WIDTH, ANGLE_WIDTH)
    B.check(B.lib().bhw_atan2_device(ctypes.byref(p), dev, _stream_ptr(torch, dev), x.numel(),
                                      ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                                      ctypes.c_void_p(phi.data_ptr())))
    return phi


def win_function(win_type, i0, count, *, nphase, nwidth, device=None, **kw):
    """HLS top function swept over i (hls/windows/win_function.h:65-69): unknown win_type -> zeros
    (win_empty, hls/windows/win_function.cpp:159-165,417-419)."""
    torch = _torch()
    if win_type not in (1, 2, 3, 4, 5, 7):
        return torch.zeros(int(count), dtype=torch.int32, device="cuda" if device is None else device)
    p = B.make_params(win_type, nphase, nwidth, model=B.MODEL_HLS, combine=B.COMBINE_HLS, **kw)
    return generate(p, i0, count, device=device)


class WinSelector:
    """entity win_selector (src/win_selector.vhd:60-87).

    Generics become constructor arguments with the reference's names; the AA0..AA6 ports are the
    `aa` list (integer, caller-scaled; None = the HLS model's built-in constants).  `model`/`combine`
    choose which of the reference's bit-models to reproduce (default: the HLS C++ model).
    """

    def __init__(self, PHI_WIDTH=10, DAT_WIDTH=16, WIN_TYPE="HAMMING", SIN_TYPE="CORDIC", LUT_SIZE=9,
                 XSERIES="ULTRA", aa=None, model=B.MODEL_HLS, combine=B.COMBINE_HLS, precision=1, device=None):
        if WIN_TYPE not in _WIN_TYPES:
            raise ValueError(f"WIN_TYPE {WIN_TYPE!r}: expected one of {sorted(_WIN_TYPES)}")
        if SIN_TYPE not in ("CORDIC", "TAYLOR", "TAYLOR_ALL"):
            raise ValueError("SIN_TYPE must be 'CORDIC' or 'TAYLOR' (or this library's extension 'TAYLOR_ALL')")
        if XSERIES not in ("7SERIES", "ULTRA"):  # selects DSP48 port widths only (tay1_order.vhd:538-578)
            raise ValueError("XSERIES must be '7SERIES' or 'ULTRA'")
        self.device = device
        self.params = B.make_params(
            _WIN_TYPES[WIN_TYPE], PHI_WIDTH, DAT_WIDTH, model=model, combine=combine,
            sin_type=self._sin_type(SIN_TYPE, _WIN_TYPES[WIN_TYPE]),
            precision=precision, lut_size=LUT_SIZE, aa=aa)
        self._phase = 0  # the PHI_WIDTH-bit counter (RESET clears it: bh_win_7term.vhd:179-186)

    @staticmethod
    def _sin_type(SIN_TYPE, win_type):
        """The selector hands SIN_TYPE only to hamming_win and bh_win_3term (src/win_selector.vhd:93-135); the 4/5/7-term
        entities have no such generic (:137-199), so "TAYLOR" there still elaborates the CORDIC design.  "TAYLOR_ALL" is
        the extension of include/bhw.h (Taylor source for every term count)."""
        if SIN_TYPE == "TAYLOR_ALL":
            return B.SIN_TAYLOR_ALL
        if SIN_TYPE == "TAYLOR" and win_type in (1, 2, 3):
            return B.SIN_TAYLOR
        return B.SIN_CORDIC

    @property
    def length(self):
        return 1 << self.params.phi_width

    def reset(self):
        self._phase = 0

    def enable(self, count, out=None, algo=B.ALGO_AUTO):
        """ENABLE high for `count` clocks: the next `count` values of DT_WIN; the counter advances and wraps."""
        w = generate(self.params, self._phase, count, device=self.device, out=out, algo=algo)
        self._phase = (self._phase + int(count)) % self.length
        return w

    def apply(self, x, shift=None, out=None):
        """The multiplier stage behind DT_WIN for the next x.numel() clocks: y = (x * DT_WIN) >> shift."""
        y = apply(self.params, x, n0=self._phase, shift=shift, out=out)
        self._phase = (self._phase + x.numel()) % self.length
        return y

    def window(self, out=None, algo=B.ALGO_AUTO):
        """One full period from phase 0."""
        return generate(self.params, 0, self.length, device=self.device, out=out, algo=algo)

    def shard(self, rank, world_size, out=None, algo=B.ALGO_AUTO, layout="contiguous"):
        """This rank's share of ONE window over `world_size` devices, no collective.
        layout "contiguous": the index range shard_range(N, rank, world_size), returned as a tensor of that length.
        layout "interleaved": the ownership part (rank, world_size) of include/bhw.h -- ring lanes with their eight quadrant /
        half-period images, so the folds still share CORDIC work; written into `out`, a full-length window buffer (allocated
        when None: elements this rank does not own stay uninitialised), which is returned; self.segments(...) lists them."""
        if layout == "contiguous":
            n0, count = shard_range(self.length, rank, world_size)
            return generate(self.params, n0, count, device=self.device, out=out, algo=algo)
        if layout != "interleaved":
            raise ValueError("layout must be 'contiguous' or 'interleaved'")
        if out is None:
            torch = _torch()
            out = torch.empty(self.length, dtype=torch.int32, device=f"cuda:{_dev_index(torch, self.device)}")
        return generate_part(self.params, rank, world_size, out, algo=algo)

    def segments(self, rank, world_size):
        """[(n0, count), ...] owned by `rank` in the interleaved layout (host arithmetic, no GPU needed)."""
        return B.part_segments(self.params, rank, world_size)
