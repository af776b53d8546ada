"""Quality harness: side-lobe levels of the generated integer windows vs the table in the
reference's README.md:30-41, with the spectrum estimate of math/window_test.m:122-161 restated in NumPy."""
import numpy as np
import pytest

from blackman_harris_win_amd import binding as B

pytestmark = pytest.mark.gpu


def sidelobe_db(w, pad=16):
    w = np.asarray(w, dtype=np.float64)
    spec = np.abs(np.fft.rfft(w, pad * len(w)))
    spec /= spec[0]
    i = 1
    while i + 1 < len(spec) and spec[i + 1] < spec[i]:      # walk down the main lobe to its first null
        i += 1
    return 20 * np.log10(spec[i:].max())


# (name, win_type, library preset or None for the HLS built-ins, README level in dB, tolerance).  The named sets come from the library
# (bhw_coeffs_preset); its Nuttall a2 is the published 0.144232 -- the comment at hls/windows/win_function.cpp:244 prints 0.144323,
# an upstream slip (with it the four weights sum to 1.000091 and the side lobes sit near -60 dB, not the README's -93).
CASES = [
    ("Hamming (25/46)", 1, None, -43, 1.5),   # measured -41.7 dB with the HLS constant 0.5434783
    ("Hann", 2, None, -32, 1.0),
    ("Blackman", 3, "blackman", -58, 1.0),
    ("Blackman-Harris 3-term", 3, "bh3", -71, 1.5),
    ("Nuttall", 4, "nuttall", -93, 1.5),
    ("Blackman-Harris 4-term", 4, None, -92, 1.0),
    ("Blackman-Nuttall", 4, "blackman-nuttall", -98, 1.0),
    ("Blackman-Harris 5-term", 5, None, -124, 2.5),
]


@pytest.mark.parametrize("name,win,preset,level,tol", CASES)
def test_sidelobe_levels_match_reference_readme(name, win, preset, level, tol):
    import blackman_harris_win_amd as bhw
    pw, w = 12, 32
    coefs = None
    if preset is not None:
        wt, coefs, _ = B.coeffs_preset(preset, w)
        assert wt == win
    # The HLS scaling of 2/3/4-term windows, round(a_k (2^(W-1)-1)), leaves no headroom: the peak a0+a1+... plus the
    # CORDIC overshoot exceeds 2^(W-1)-1 and the win_t store wraps it (faithfully reproduced; e.g. Hann 10/24 in
    # tests/test_oracle.py).  A consumer drives the AA ports one bit lower, as done here.
    aa = B.coeffs_from_float(win, w - 1 if win < 5 else w, coefs)
    p = B.make_params(win, pw, w, aa=aa)
    got = sidelobe_db(bhw.generate(p, 0, 1 << pw).cpu().numpy())
    assert abs(got - level) <= tol, (name, got, level)


def test_bh7_reaches_180db_class():
    """README.md:45-53: the 7-term set 'gives you up to 180 dB side lobe level' -- needs the full 32-bit width."""
    import blackman_harris_win_amd as bhw
    wt, readme, aa32 = B.coeffs_preset("bh7-readme", 32)
    assert wt == 7 and aa32 == B.coeffs_from_float(7, 32, readme)
    p = B.make_params(7, 12, 32, aa=aa32)
    lvl32 = sidelobe_db(bhw.generate(p, 0, 1 << 12).cpu().numpy())
    assert lvl32 < -160
    # "1 digital bit equals 6 dB" (README.md:5-6): at 16 bits the quantisation floor dominates
    p16 = B.make_params(7, 12, 16, aa=B.coeffs_from_float(7, 16, readme))
    lvl16 = sidelobe_db(bhw.generate(p16, 0, 1 << 12).cpu().numpy())
    assert -110 < lvl16 < -70 and lvl16 > lvl32 + 50
