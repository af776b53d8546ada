"""blackman_harris_win_amd -- MI355X-native fixed-point window-coefficient generator.

Python host side above the C ABI of include/bhw.h.  PyTorch supplies device memory, streams and
torch.distributed only; every coefficient is computed by the hand-written HIP kernels in
libbhw.so.  There is no CPU compute path here: if the library is missing or no GPU is present,
the compute calls raise.

The surface mirrors the reference's operator:
  * ``WinSelector``  <->  entity win_selector (src/win_selector.vhd:60-87): generics PHI_WIDTH,
    DAT_WIDTH, WIN_TYPE, SIN_TYPE, LUT_SIZE and weight ports AA0..AA6; ``enable(count)`` is
    "hold ENABLE high for count clocks" and returns DT_WIN.
  * ``win_function(win_type, i0, count, ...)``  <->  HLS top win_function(win_type, i, &out)
    (hls/windows/win_function.h:65-69) swept over i.
  * ``cordic(theta0, count, ...)``  <->  cordic() (cpp/cordic_sincos.cpp:10, hls/cordic/cordic.cpp:45).
"""
from .binding import (  # noqa: F401
    ALGO_AUTO, ALGO_DIRECT, ALGO_FUSED, ALGO_TABLE,
    TABLE_BEST, TABLE_DELTA16, TABLE_NIBBLE, TABLE_NIBBLE_ESC, TABLE_PLAIN, TABLE_RESIDUAL,
    COMBINE_HLS, COMBINE_VHDL,
    MODEL_CPP, MODEL_DDS48, MODEL_HLS, MODEL_SCALED, MODEL_VHDL,
    SIN_CORDIC, SIN_TAYLOR, SIN_TAYLOR_ALL,
    WIN_BH3, WIN_BH4, WIN_BH5, WIN_BH7, WIN_HAMMING, WIN_HANN,
    BhwAtan2Params, BhwError, BhwParams, coeffs_from_float, constant_tables, lib, lib_path, make_params, part_segments,
)
from .selector import (  # noqa: F401
    WinSelector, apply, atan2, cordic, gather_parts, generate, generate_batched, generate_part, prepare, shard_range, win_function,
)

__all__ = [
    "WinSelector", "win_function", "cordic", "atan2", "generate", "generate_batched", "generate_part", "gather_parts", "part_segments", "apply",
    "prepare", "shard_range",
    "make_params", "coeffs_from_float", "constant_tables", "BhwParams", "BhwError", "lib", "lib_path",
]
