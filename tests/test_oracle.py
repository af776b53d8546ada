"""Oracle pinning (CPU): the restatement vs the reference's own outputs, golden vectors and tests."""
import hashlib
import math
import os

import numpy as np
import pytest

import oracle_lib as O

REFERENCE = "/root/reference"


def _md5(a):
    return hashlib.md5(np.ascontiguousarray(a, dtype="<i4").tobytes()).hexdigest()


# ---- constants ------------------------------------------------------------------------------------
def test_tables_anchors():
    t2, t4, g46, g47 = O.tables()
    assert t2[0] == 0x200000000000 and t2[1] == 0x12E4051D9DF3 and t2[47] == 0
    assert t4[0] == 0x400000000000 and t4[1] == 0x25C80A3B3BE6 and t4[47] == 0
    assert g46 == 0x26DD3B6A10D8 and g47 == 0x4DBA76D421AF


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="upstream checkout not present")
def test_tables_equal_reference_literals():
    """Closed-form tables == the 48 literals in the reference sources (read as text, data only)."""
    import re
    t2, t4, _, _ = O.tables()
    txt = open(os.path.join(REFERENCE, "cpp", "cordic_sincos.cpp")).read()
    lits = [int(x, 16) for x in re.findall(r"0x([0-9A-Fa-f]{12})\b", txt.split("lut_table")[1].split("};")[0])]
    assert lits == t2
    txt = open(os.path.join(REFERENCE, "hls", "windows", "win_function.cpp")).read()
    lits = [int(x, 16) for x in re.findall(r"0x([0-9A-Fa-f]{12})\b", txt.split("lut_table")[1].split("};")[0])]
    assert lits == t4
    txt = open(os.path.join(REFERENCE, "src", "cordic_dds.vhd")).read()
    lits = [int(x, 16) for x in re.findall(r'x"([0-9A-Fa-f]{12})"', txt.split("ROM_LUT : rom_array")[1].split(");")[0])]
    assert lits == t4


# ---- model A pinned by the reference itself -----------------------------------------------------------
def test_model_a_equals_reference_coe_dat(golden, golden_dir):
    e = golden["coe_cpp_14_12"]
    sc = np.load(os.path.join(golden_dir, e["file"])).astype(np.int32)
    s, c = O.sincos(O.oparams(1, 14, 12, model=O.MODEL_CPP), 0, 1 << 14)
    assert np.array_equal(s, sc[:, 0]) and np.array_equal(c, sc[:, 1])
    text = "".join("%d %d\n" % (a, b) for a, b in zip(s, c)).encode()   # cpp/cordic_sincos.cpp:138
    assert hashlib.md5(text).hexdigest() == e["text_md5"] == "b65f091fb2afeeb252aa0bc5728fe46a"
    assert s.min() == -1025 and s.max() == 1024 and c.min() == -1025 and c.max() == 1024


def test_model_a_equals_reference_vectors(golden, golden_dir):
    names = [k for k in golden if k.startswith("sincos_cpp_")]
    assert len(names) >= 10
    for name in names:
        th, s, c = np.load(os.path.join(golden_dir, golden[name]["file"]))
        pr = golden[name]["params"]
        p = O.oparams(1, pr["phi_width"], pr["dat_width"], model=O.MODEL_CPP)
        for i in range(0, len(th), max(1, len(th) // 600)):
            so, co = O.sincos(p, int(th[i]), 1)
            assert (so[0], co[0]) == (s[i], c[i]), (name, int(th[i]))


@pytest.mark.skipif(not O.ref_pairs(), reason="oracle/_ref not built")
def test_model_a_equals_oracle_ref_live():
    """Live comparison against the reference's compiled cordic() (oracle/_ref)."""
    rng = np.random.default_rng(7)
    for pw, w, path in O.ref_pairs():
        ref = O.RefCordic(path)
        n = 1 << pw
        th = np.arange(n) if pw <= 10 else np.unique(np.concatenate([rng.integers(0, n, 400), [0, n // 4, n // 2, n - 1]]))
        s, c = ref.sweep(th)
        p = O.oparams(1, pw, w, model=O.MODEL_CPP)
        for i, t in enumerate(th):
            so, co = O.sincos(p, int(t), 1)
            assert (so[0], co[0]) == (s[i], c[i]), (pw, w, int(t))


# ---- model B pinned by SURVEY App. B known answers ------------------------------------------------------
@pytest.mark.parametrize("name", ["C1_hamming_12_16", "C2_bh4_20_24", "C4_bh4_16_24_frame", "bh5_10_24", "bh7_4_16"])
def test_model_b_known_answers(golden, golden_dir, name):
    e = golden[name]
    pr = e["params"]
    p = O.oparams(0, pr["phi_width"], pr["dat_width"], aa=pr["aa"], n_terms=pr["n_terms"])
    a = O.generate(p, e["n0"], e["count"])
    assert _md5(a) == e["md5"]
    assert int(a.astype(np.int64).sum()) == e["sum"] and int(a.min()) == e["min"] and int(a.max()) == e["max"]
    if "file" in e:
        assert np.array_equal(a, np.load(os.path.join(golden_dir, e["file"])))
    for n, v in e.get("sparse", {}).items():
        assert int(a[int(n)]) == v


def test_model_b_c3_sparse_and_strided(golden):
    e = golden["C3_bh7_26_32"]
    p = O.oparams(7, 26, 32)
    assert list(p.aa) == e["params"]["aa"] == [291220644, 465407608, 234080144, 70636474, 11555467, 826795, 14690]
    for n, v in e["sparse"].items():
        assert int(O.generate(p, int(n), 1)[0]) == v
    step = (1 << 23) // 1024
    for g in (0, 3, 7):
        got = [int(O.generate(p, (g << 23) + i * step, 1)[0]) for i in range(0, 1024, 16)]
        assert got == e["shards"][g]["strided_1024"][::16]


def test_hls_builtin_coefficients():
    # SURVEY 8d parameter sets
    assert O.coeffs(1, 16)[:2] == [17808, 14959]
    assert O.coeffs(4, 24)[:4] == [3009413, 4096073, 1185142, 97979]
    assert O.coeffs(7, 32) == [291220644, 465407608, 234080144, 70636474, 11555467, 826795, 14690]


# ---- the reference's own tolerance tests, restated ---------------------------------------------------
@pytest.mark.parametrize("model", [O.MODEL_HLS, O.MODEL_CPP, O.MODEL_VHDL])
def test_reference_cordic_tolerance_rule(model):
    """hls/cordic/cordic_test.cpp:66-93: mean |err| per channel < 10 LSB vs round(2^(W-2) sin/cos)."""
    pw, w = 10, 16
    n = 1 << pw
    s, c = O.sincos(O.oparams(1, pw, w, model=model), 0, n)
    i = np.arange(n)
    ts = np.round(2.0 ** (w - 2) * np.sin(2 * i * math.pi / n))
    tc = np.round(2.0 ** (w - 2) * np.cos(2 * i * math.pi / n))
    assert np.abs(s - ts).sum() / n < 10 and np.abs(c - tc).sum() / n < 10
    assert np.abs(s - ts).max() <= 5 and np.abs(c - tc).max() <= 5


_COEF = {1: [0.5434783, 1 - 0.5434783], 2: [0.5, 0.5], 3: [0.21, 0.25, 0.04], 4: [0.35875, 0.48829, 0.14128, 0.01168],
         5: [0.3232153788877343, 0.4714921439576260, 0.1755341299601972, 0.0284969901061499, 0.0012613570882927],
         7: [0.271220360585039, 0.433444612327442, 0.218004122892930, 0.065785343295606, 0.010761867305342,
             0.000770012710581, 0.000013680883060]}


def ideal_window(win, n):
    i = np.arange(n)
    w = np.zeros(n)
    for k, a in enumerate(_COEF[win]):
        w += ((-1) ** k) * a * np.cos(2 * k * i * math.pi / n)
    return w


@pytest.mark.parametrize("win", [1, 2, 3, 4, 5, 7])
def test_reference_window_tolerance_rule(win):
    """hls/windows/window_test.cpp:93-216: sqrt(sum err^2)/N < 10 vs round((2^(W-shift)-1) * w_float)."""
    pw, w = 10, 24
    n = 1 << pw
    a = O.generate(O.oparams(win, pw, w), 0, n).astype(np.float64)
    shift = 2 if win in (5, 7) else 1
    gold = np.round((2.0 ** (w - shift) - 1.0) * ideal_window(win, n))
    # Hann at 10/24 peaks at a0 + a1 = 2^23, which the win_t store wraps to -2^23 (faithful to the HLS
    # model; the reference's own check trips on that one sample).  Compare modulo 2^W.
    err = np.mod(a - gold + 2.0 ** (w - 1), 2.0 ** w) - 2.0 ** (w - 1)
    assert math.sqrt((err ** 2).sum()) / n < 10
    assert np.abs(err).max() <= 4
    if win != 2:
        assert np.array_equal(err, a - gold)


def test_vhdl_rule_matches_ideal_half_amplitude():
    """with the CORDIC source the VHDL rule yields (A0 - A1/2 cos)/2 for Hamming."""
    pw, w = 11, 16
    n = 1 << pw
    a0, a1 = round(0.5434783 * (2 ** (w - 1) - 1)), round((1 - 0.5434783) * (2 ** (w - 1) - 1))
    got = O.generate(O.oparams(1, pw, w, model=O.MODEL_VHDL, combine=O.COMBINE_VHDL, aa=[a0, a1]), 0, n)
    ideal = (a0 - 0.5 * a1 * np.cos(2 * math.pi * np.arange(n) / n)) / 2
    assert np.abs(got - ideal).max() < 3


def test_taylor_close_to_float():
    for pw, w, L in [(12, 16, 9), (14, 24, 9), (16, 32, 9), (10, 16, 9), (11, 16, 9)]:
        n = 1 << pw
        p = O.oparams(1, pw, w, sin_type=O.SIN_TAYLOR, lut_size=L)
        s, c = O.sincos(p, 0, n)
        amp = 2.0 ** (w - 1) - 1
        i = np.arange(n)
        tol = 4 + amp * (math.pi / 2 ** (L + 1)) ** 2 / 2 * 1.1 if pw - L > 2 else 2
        assert np.abs(s - amp * np.sin(2 * math.pi * i / n)).max() <= tol
        assert np.abs(c - amp * np.cos(2 * math.pi * i / n)).max() <= tol


def test_taylor_all_extension():
    """BHW_SIN_TAYLOR_ALL (include/bhw.h): identical to the reference wiring for 2/3 terms; for 4/5/7 terms the harmonic
    k = m*2^v comes from the PHASE_WIDTH-v generator, and the window stays within a few LSB of the ideal one."""
    for win, pw, w in [(1, 11, 16), (3, 12, 24)]:
        a = O.generate(O.oparams(win, pw, w, sin_type=O.SIN_TAYLOR, lut_size=9), 0, 1 << pw)
        b = O.generate(O.oparams(win, pw, w, sin_type=O.SIN_TAYLOR_ALL, lut_size=9), 0, 1 << pw)
        assert np.array_equal(a, b)
    coe = {4: [0.35875, 0.48829, 0.14128, 0.01168],
           7: [0.271220360585039, 0.433444612327442, 0.218004122892930, 0.065785343295606, 0.010761867305342,
               0.000770012710581, 0.000013680883060]}
    for win, pw, w in [(4, 12, 16), (7, 13, 24), (7, 12, 32)]:
        n = 1 << pw
        p = O.oparams(win, pw, w, sin_type=O.SIN_TAYLOR_ALL, lut_size=9)
        got = O.generate(p, 0, n).astype(np.float64)
        x = 2 * math.pi * np.arange(n) / n
        # weights a_k scale to 2^(W-s)-1, cosines to 2^(W-1)-1, products >> (W-2): unit = 2 * a_k * cos
        amp = 2.0 ** (w - 1) - 1
        ideal = sum((-1) ** k * p.aa[k] * (np.cos(k * x) * amp if k else 2.0 ** (w - 2)) for k in range(len(coe[win]))) / 2.0 ** (w - 2)
        # 1st-order Taylor error per harmonic ~ amp * (pi / 2^(L+1))^2 / 2 relative to full scale, plus truncations
        tol = 8 + sum(abs(p.aa[k]) for k in range(1, len(coe[win]))) / 2.0 ** (w - 2) * (amp * (math.pi / 2 ** 10) ** 2 / 2 + 2)
        err = (got - ideal + 2.0 ** (w - 1)) % 2.0 ** w - 2.0 ** (w - 1)      # s = 1 windows wrap at their peak, like the HLS model
        assert np.abs(err).max() <= tol, (win, pw, w, np.abs(err).max(), tol)
        # the 4th harmonic really is the PHASE_WIDTH-2 generator: cos of harmonic 4 at n equals generator(PW-2) at n mod N/4
        c4, _ = O.taylor(pw - 2, w, 9, np.arange(n) % (n // 4))
        c1, _ = O.taylor(pw, w, 9, (4 * np.arange(n)) % n)
        assert np.abs(c4.astype(np.int64) - c1).max() <= 8      # same angle, different rounding path


# ---- variant generators: restated from source, no upstream vectors -> sanity against the ideal functions
@pytest.mark.parametrize("model", [O.MODEL_DDS48, O.MODEL_SCALED])
def test_variant_generators_close_to_float(model):
    """cordic_dds48 / cordic_dds_scaled as written: DT_COS = +cos, DT_SIN = -sin at amplitude 2^(DATA_WIDTH-2)."""
    for pw, w in [(10, 16), (12, 12), (14, 24), (13, 32), (16, 8), (9, 20)]:
        n = 1 << pw
        th = np.arange(n) if pw <= 12 else np.random.default_rng(pw).integers(0, n, 3000)
        p = O.oparams(1, pw, w, model=model)
        s = np.empty(len(th), np.int32)
        c = np.empty(len(th), np.int32)
        for i, t in enumerate(th):
            s1, c1 = O.sincos(p, int(t), 1)
            s[i], c[i] = s1[0], c1[0]
        x = 2 * math.pi * th / n
        amp = 2.0 ** (w - 2)
        assert np.abs(c - amp * np.cos(x)).max() < 3 and np.abs(s + amp * np.sin(x)).max() < 3
    with pytest.raises(ValueError):                                   # no window entity instantiates them
        O.generate(O.oparams(4, 10, 16, model=model), 0, 4)


def test_variant_goldens_reproduce(golden, golden_dir):
    """The committed vectors of the variant generators and of the Taylor extension are what the oracle says today."""
    import hashlib
    md5 = lambda a: hashlib.md5(np.ascontiguousarray(a, dtype="<i4").tobytes()).hexdigest()
    for name, e in golden.items():
        if name.startswith("atan2_"):
            x, y, phi = np.load(os.path.join(golden_dir, e["file"]))
            assert np.array_equal(O.atan2(e["precision"], e["input_width"], e["angle_width"], x, y), phi)
        elif name.startswith("sincos_dds48_") or name.startswith("sincos_scaled_"):
            pr = e["params"]
            s, c = O.sincos(O.oparams(1, pr["phi_width"], pr["dat_width"], model=pr["model"]), e["theta0"], e["count"])
            assert md5(s) == e["sin_md5"] and md5(c) == e["cos_md5"]
        elif name.startswith("taylor_all_"):
            pr = e["params"]
            p = O.oparams({4: 4, 5: 5, 7: 7}[pr["n_terms"]], pr["phi_width"], pr["dat_width"], combine=pr["combine"],
                          sin_type=pr["sin_type"], lut_size=pr["lut_size"], aa=pr["aa"])
            assert md5(O.generate(p, e["n0"], e["count"])) == e["md5"]


def test_variant_generator_wraps_never_fire():
    """The SIZE- / DWPH-bit stores of cordic_dds48 and cordic_dds_scaled never change a value (k_sincos_prerot omits them)."""
    import ctypes
    o = O.oracle()
    ev = ctypes.c_uint64(0)
    c = np.empty(1, np.int32)
    s = np.empty(1, np.int32)
    rng = np.random.default_rng(11)
    for model in (O.MODEL_DDS48, O.MODEL_SCALED):
        for pw, w in [(10, 8), (10, 16), (12, 12), (18, 16), (20, 24), (26, 32), (30, 9), (30, 30), (9, 32), (32, 21), (4, 10)]:
            n = 1 << pw
            th = np.arange(n) if pw <= 12 else np.unique(np.concatenate(
                [rng.integers(0, n, 2500), np.arange(64), n - 1 - np.arange(64)] +
                [q * (n // 4) + np.arange(-64, 64) for q in range(1, 4)] + [n // 8 + np.arange(-8, 8)]) % n)
            for t in th:
                assert o.bhwo_cordic(model, pw, w, 1, int(t), c.ctypes.data, s.ctypes.data, ctypes.byref(ev)) == 0
    assert ev.value == 0


def test_dds48_is_the_48_bit_case_of_scaled():
    """SEL_SIZE(DATA_WIDTH-8) = 48 from DATA_WIDTH = 29 on (src/cordic_dds_scaled.vhd:102-107): both entities agree there."""
    for w in (29, 30, 32):
        a = O.sincos(O.oparams(1, 12, w, model=O.MODEL_DDS48), 0, 4096)
        b = O.sincos(O.oparams(1, 12, w, model=O.MODEL_SCALED), 0, 4096)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    a = O.sincos(O.oparams(1, 12, 16, model=O.MODEL_DDS48), 0, 4096)
    b = O.sincos(O.oparams(1, 12, 16, model=O.MODEL_SCALED), 0, 4096)
    assert not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]))   # 30-bit data path rounds differently


def test_atan2_as_written():
    """cordic_atan2 as written (src/cordic_atan2.vhd:207-213): z accumulates -atan(|y|/|x|), PHI_PI is a quarter turn."""
    rng = np.random.default_rng(5)
    for P, IW, AW in [(1, 23, 24), (3, 16, 16), (4, 32, 32), (2, 15, 16)]:
        lim = 1 << (min(IW, AW - 1) - 1)
        x = rng.integers(lim // 4, lim, 600) * rng.choice([-1, 1], 600)      # well-conditioned magnitudes
        y = rng.integers(lim // 4, lim, 600) * rng.choice([-1, 1], 600)
        got = O.atan2(P, IW, AW, x, y).astype(np.float64)
        phi0 = np.arctan2(np.abs(y), np.abs(x)) / (2 * math.pi) * 2.0 ** AW
        quarter = 2.0 ** AW / 4
        want = np.where((x >= 0) & (y >= 0), -phi0, np.where((x >= 0) & (y < 0), -phi0 + quarter,
                        np.where((x < 0) & (y >= 0), phi0, -phi0 - quarter)))
        d = (got - want + 2.0 ** (AW - 1)) % 2.0 ** AW - 2.0 ** (AW - 1)
        assert np.abs(d).max() < 2.0 ** AW * 4 / lim + 8, (P, IW, AW, np.abs(d).max())
    with pytest.raises(ValueError):
        O.atan2(1, 20, 24, [1], [1])        # upstream default generics: VEC_DX(22) does not exist


# ---- structural facts the kernels rely on -----------------------------------------------------------
def test_typed_store_wraps_never_fire():
    """The W+2 / W+P bit wraps of the HLS/VHDL CORDIC never change a value (kernels omit them)."""
    import ctypes
    o = O.oracle()
    ev = ctypes.c_uint64(0)
    c = np.empty(1, np.int32)
    s = np.empty(1, np.int32)
    rng = np.random.default_rng(3)
    for model, prec in [(O.MODEL_HLS, 1), (O.MODEL_VHDL, 1), (O.MODEL_VHDL, 4)]:
        for pw, w in [(10, 8), (10, 16), (12, 12), (18, 16), (20, 24), (26, 32), (26, 24), (30, 30), (9, 32)]:
            if model == O.MODEL_HLS and pw > w + 2:
                continue
            n = 1 << pw
            th = np.arange(n) if pw <= 12 else np.unique(np.concatenate(
                [rng.integers(0, n, 3000), np.arange(64), n // 4 + np.arange(-64, 64), n // 8 + np.arange(-8, 8)]) % n)
            for t in th:
                assert o.bhwo_cordic(model, pw, w, prec, int(t), c.ctypes.data, s.ctypes.data, ctypes.byref(ev)) == 0
    assert ev.value == 0


def test_residual_format_margin():
    """The 2-byte "residual" table format (csrc/bhw_device.h) stores (c, s) minus a straight line through exact records 2^d
    entries apart in one signed byte per component.  Measure that deviation over every model, at the widths the format is used
    for (d from the same rule as bhwk_resid_dlog): it must stay far inside int8."""
    rng = np.random.default_rng(17)
    worst = 0
    for model, prec, out_shr in [(O.MODEL_HLS, 1, 2), (O.MODEL_CPP, 1, 2), (O.MODEL_VHDL, 1, 1), (O.MODEL_VHDL, 2, 2)]:
        for pw, w in [(26, 32), (24, 32), (22, 30), (22, 24), (20, 24), (26, 26), (23, 31), (28, 32), (30, 32), (21, 27)]:
            if model == O.MODEL_HLS and pw > w + 2:
                continue
            if pw >= w or w + out_shr > 34:                       # phase bits dropped / 64-bit build: plain table
                continue
            d = min(9, (2 * pw - (w - 2) - 4) // 2)
            if d <= 6:
                continue
            E = 1 << (pw - 2)
            p = O.oparams(1, pw, w, model=model, precision=prec)
            cells = np.unique(np.concatenate([[0, 1, (E >> d) - 2, (E >> d) // 2], rng.integers(0, (E >> d) - 1, 3)]))
            for cell in cells:
                u0 = int(cell) << d
                s, c = O.sincos(p, u0, (1 << d) + 1)
                f = np.arange(1 << d, dtype=np.int64)
                pc = c[0] + (((int(c[-1]) - int(c[0])) * f) >> d)
                ps = s[0] + (((int(s[-1]) - int(s[0])) * f) >> d)
                worst = max(worst, int(np.abs(c[:-1] - pc).max()), int(np.abs(s[:-1] - ps).max()))
    assert 0 < worst <= 40, worst


def test_nibble_escape_capacity():
    """The "nibble + escapes" table format (csrc/bhw_device.h, format 5): one byte per entry, the deviation from the residual format's
    straight line in two 4-bit fields, the low field -8 reserved as the marker of an entry listed exactly in the hash table of the
    build workgroup that stores it (128 slots, at most 96 used; a workgroup owns 2^14 entries of [0, E/2) and their images E - u for
    tables of 2^24 entries).  The records of this format carry c + 1, s + 1 (kEscBias: the deviations lean positive).  Measured here
    over the WHOLE table of the headline window for the two models whose noise is wider than the fields: 547 / 932 listed entries of
    2^24 (3 .. 6 per 100 000; 1 005 / 1 486 without the bias), the fullest workgroup inside its table."""
    pw, w, d = 26, 32, 9
    E = 1 << (pw - 2)
    own = 1 << 14
    for model, lo, hi in ((O.MODEL_CPP, 450, 650), (O.MODEL_VHDL, 800, 1100)):
        s, c = O.sincos(O.oparams(1, pw, w, model=model), 0, E + 1)
        s, c = s.astype(np.int64), c.astype(np.int64)
        f = np.arange(1 << d, dtype=np.int64)
        hc, hs = c[::1 << d], s[::1 << d]
        dc = c[:E].reshape(-1, 1 << d) - (hc[:-1, None] + (((hc[1:] - hc[:-1])[:, None] * f) >> d))
        ds = s[:E].reshape(-1, 1 << d) - (hs[:-1, None] + (((hs[1:] - hs[:-1])[:, None] * f) >> d))
        assert max(int(np.abs(dc).max()), int(np.abs(ds).max())) > 8                       # four-bit fields alone do not hold it
        dc, ds = dc - 1, ds - 1                                                            # kEscBias
        esc = ((dc < -7) | (dc > 7) | (ds < -8) | (ds > 7)).reshape(-1)                    # dc == -8 is the marker itself
        total = int(esc.sum())
        assert lo <= total <= hi, (model, total)
        per_own = esc[:E // 2].reshape(-1, own).sum(axis=1)
        per_img = esc[E // 2 + 1:][::-1].reshape(-1)[:E // 2 - 1]                         # image E - u of source u = 1 .. E/2 - 1
        per_img = np.concatenate([[0], per_img]).reshape(-1, own).sum(axis=1)
        per_wg = per_own + per_img
        per_wg[-1] += int(esc[E // 2])                                                     # the middle entry: the last workgroup
        assert int(per_wg.sum()) == total and int(per_wg.max()) <= 96, (model, int(per_wg.max()))


def test_quadrant_images_share_first_quadrant_result():
    """cos/sin at theta + j*N/4 are the quadrant-rotated first-quadrant pair (basis of the table strategy)."""
    for model in (O.MODEL_HLS, O.MODEL_CPP, O.MODEL_VHDL):
        pw, w = 12, 20
        n = 1 << pw
        s, c = O.sincos(O.oparams(1, pw, w, model=model), 0, n)
        q = n // 4
        neg = (lambda v: ~v) if model == O.MODEL_CPP else (lambda v: -v)
        assert np.array_equal(c[q:2 * q], neg(s[:q])) and np.array_equal(s[q:2 * q], c[:q])
        assert np.array_equal(c[2 * q:3 * q], neg(c[:q])) and np.array_equal(s[2 * q:3 * q], neg(s[:q]))
        assert np.array_equal(c[3 * q:], s[:q]) and np.array_equal(s[3 * q:], neg(c[:q]))


def test_stream_is_periodic():
    p = O.oparams(4, 8, 16)
    a = O.generate(p, 0, 256)
    assert np.array_equal(O.generate(p, 256 * 5 + 17, 100), np.concatenate([a, a])[17:117])


def test_oracle_rejects_bad_params():
    with pytest.raises(ValueError):
        O.generate(O.oparams(4, 26, 16), 0, 4)        # HLS model: PW > W + 2
    with pytest.raises(ValueError):
        O.generate(O.oparams(4, 10, 16, n_terms=6), 0, 4)
    with pytest.raises(ValueError):
        O.generate(O.oparams(5, 10, 16, sin_type=O.SIN_TAYLOR), 0, 4)   # reference wiring: 2-/3-term only
    assert len(O.generate(O.oparams(5, 10, 16, sin_type=O.SIN_TAYLOR_ALL), 0, 4)) == 4
