"""Parity proper (GPU): every result comes through the C ABI (libbhw.so HIP kernels) and is compared
bit-for-bit with the oracle on the same parameters, with the committed golden vectors, and -- at
BASELINE.json's full sizes -- through checksums and size-independent properties."""
import ctypes
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as O
from blackman_harris_win_amd import binding as B

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _md5(a):
    return hashlib.md5(np.ascontiguousarray(a, dtype="<i4").tobytes()).hexdigest()


def gpu_generate(p, n0, count, algo=B.ALGO_AUTO):
    import blackman_harris_win_amd as bhw
    return bhw.generate(p, n0, count, algo=algo).cpu().numpy()


def params_from_golden(pr):
    return B.make_params(pr.get("win_type", 0) or {2: 1, 3: 3, 4: 4, 5: 5, 7: 7}[pr["n_terms"]], pr["phi_width"], pr["dat_width"],
                         model=pr["model"], combine=pr["combine"], sin_type=pr["sin_type"], precision=pr["precision"],
                         lut_size=pr["lut_size"], aa=pr["aa"], n_terms=pr["n_terms"])


ALGOS = [B.ALGO_DIRECT, B.ALGO_TABLE]


# ---- committed golden vectors --------------------------------------------------------------------------
@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("name", ["C1_hamming_12_16", "C2_bh4_20_24", "C4_bh4_16_24_frame", "bh5_10_24", "bh7_4_16",
                                  "vhdl_hamming_11_16", "vhdl_bh7_10_24", "vhdl_bh4_12_24_p3", "cpp_bh7_12_32_hlscombine"])
def test_golden_windows(torch, golden, golden_dir, name, algo):
    e = golden[name]
    got = gpu_generate(params_from_golden(e["params"]), e["n0"], e["count"], algo)
    assert _md5(got) == e["md5"]
    if "file" in e:
        assert np.array_equal(got, np.load(os.path.join(golden_dir, e["file"])))
    assert O.fnv(got) == e["fnv1a64"]


@pytest.mark.parametrize("name", ["taylor_hamming_12_16_l9", "taylor_bh3_14_24_l9", "taylor_all_bh7_12_24_l9",
                                  "taylor_all_bh5_13_16_l9", "taylor_all_bh4_14_32_l10"])
def test_golden_taylor(torch, golden, golden_dir, name):
    e = golden[name]
    got = gpu_generate(params_from_golden(e["params"]), e["n0"], e["count"])
    assert _md5(got) == e["md5"]


def test_golden_variant_generators(torch, golden, golden_dir):
    """Committed vectors of cordic_dds48 / cordic_dds_scaled / cordic_atan2 (oracle-made: parity unpinned)."""
    import blackman_harris_win_amd as bhw
    names = [k for k in golden if k.startswith("sincos_dds48_") or k.startswith("sincos_scaled_")]
    assert len(names) >= 5
    for name in names:
        e = golden[name]
        pr = e["params"]
        p = B.make_params(1, pr["phi_width"], pr["dat_width"], model=pr["model"])
        s, c = bhw.cordic(p, e["theta0"], e["count"])
        assert _md5(s.cpu().numpy()) == e["sin_md5"] and _md5(c.cpu().numpy()) == e["cos_md5"], name
    names = [k for k in golden if k.startswith("atan2_")]
    assert len(names) >= 3
    for name in names:
        e = golden[name]
        x, y, phi = np.load(os.path.join(golden_dir, e["file"]))
        got = bhw.atan2(torch.tensor(x.astype(np.int32), device="cuda"), torch.tensor(y.astype(np.int32), device="cuda"),
                        PRECISION=e["precision"], INPUT_WIDTH=e["input_width"], ANGLE_WIDTH=e["angle_width"])
        assert np.array_equal(got.cpu().numpy(), phi.astype(np.int32)), name


# Taylor source: reference wiring (2-/3-term) and the all-term-count extension (BHW_SIN_TAYLOR_ALL, include/bhw.h).
# (win, pw, w, lut_size): generator modes PW-L < 2 / == 2 / > 2 for every generator width PW-v in use, narrow (W < 19)
# and wide rounding variants, int32 fast path (W <= 16), ROM in LDS and beyond it (L = 13).
TAYLOR_CASES = [(1, 12, 16, 9), (2, 10, 16, 9), (3, 14, 24, 9), (3, 11, 16, 9), (3, 12, 18, 10), (1, 8, 12, 9),
                (3, 10, 20, 9), (1, 16, 32, 9), (3, 16, 31, 11), (3, 17, 16, 13), (1, 18, 24, 4),
                (4, 12, 16, 9), (4, 14, 24, 9), (5, 13, 16, 9), (5, 15, 32, 9), (7, 14, 16, 9), (7, 16, 32, 9),
                (7, 12, 24, 9), (7, 11, 30, 9), (7, 13, 18, 10), (5, 10, 20, 9), (7, 17, 16, 13), (4, 18, 30, 12),
                (7, 5, 16, 2), (5, 6, 12, 3), (7, 18, 19, 6)]


@pytest.mark.parametrize("combine", [B.COMBINE_HLS, B.COMBINE_VHDL])
@pytest.mark.parametrize("win,pw,w,L", TAYLOR_CASES)
def test_taylor_window_matches_oracle(torch, win, pw, w, L, combine):
    sin_type = B.SIN_TAYLOR if win <= 3 else B.SIN_TAYLOR_ALL
    p = B.make_params(win, pw, w, combine=combine, sin_type=sin_type, lut_size=L)
    n = 1 << pw
    op = O.from_bhw(p)
    # whole period(s) + ragged ends: fold kernel for the periods, general kernel for head and tail
    n0, count = (n - 37, 2 * n + 91) if pw <= 14 else (0, n)
    got = gpu_generate(p, n0, count)
    if pw <= 14:
        assert np.array_equal(got, O.generate(op, n0, count))
    else:
        rng = np.random.default_rng(pw * 100 + w)
        idx = np.unique(np.concatenate([np.arange(300), n // 4 + np.arange(-150, 150), n // 2 + np.arange(-150, 150),
                                        3 * (n // 4) + np.arange(-150, 150), n - 1 - np.arange(300),
                                        n // 8 + np.arange(-50, 50), n // 16 * 3 + np.arange(-50, 50),
                                        rng.integers(0, n, 1500)]))
        want = np.array([O.generate(op, int(i), 1)[0] for i in idx], dtype=np.int32)
        assert np.array_equal(got[idx], want)
        # ragged call into the middle of the period: general kernel
        mid = gpu_generate(p, n // 3, 1000)
        assert np.array_equal(mid, got[n // 3:n // 3 + 1000])


def test_taylor_all_equals_reference_wiring_for_2_and_3_terms(torch):
    for win, pw, w in [(1, 13, 16), (3, 13, 24), (3, 16, 32)]:
        a = gpu_generate(B.make_params(win, pw, w, sin_type=B.SIN_TAYLOR, lut_size=9), 0, 1 << pw)
        b = gpu_generate(B.make_params(win, pw, w, sin_type=B.SIN_TAYLOR_ALL, lut_size=9), 0, 1 << pw)
        assert np.array_equal(a, b)


def test_win_selector_ignores_taylor_for_4_5_7_terms_like_the_reference(torch):
    """src/win_selector.vhd:137-199: the BH4/5/7 entities take no SIN_TYPE, the selector elaborates CORDIC."""
    import blackman_harris_win_amd as bhw
    a = bhw.WinSelector(PHI_WIDTH=12, DAT_WIDTH=24, WIN_TYPE="BH7TERM", SIN_TYPE="TAYLOR").window()
    b = bhw.WinSelector(PHI_WIDTH=12, DAT_WIDTH=24, WIN_TYPE="BH7TERM", SIN_TYPE="CORDIC").window()
    assert bool((a == b).all())
    # the extension with the VHDL cosine-sum (full-scale cosines, halved products, final /4: bh_win_7term.vhd:353-438):
    # DT_WIN = (A0 - A1 cos x + A2 cos 2x - ...) / 4 within a few LSB
    sel = bhw.WinSelector(PHI_WIDTH=12, DAT_WIDTH=24, WIN_TYPE="BH7TERM", SIN_TYPE="TAYLOR_ALL", combine=B.COMBINE_VHDL)
    c = sel.window().cpu().numpy().astype(np.float64)
    x = 2 * np.pi * np.arange(1 << 12) / (1 << 12)
    ideal = sum((-1) ** k * sel.params.aa[k] * np.cos(k * x) for k in range(7)) / 4
    assert np.abs(c - ideal).max() < 12


@pytest.mark.parametrize("model", [B.MODEL_DDS48, B.MODEL_SCALED])
@pytest.mark.parametrize("pw,w", [(10, 16), (12, 12), (14, 24), (16, 32), (20, 8), (26, 32), (30, 20), (13, 31), (9, 29)])
def test_variant_generators_match_oracle(torch, model, pw, w):
    """cordic_dds48 / cordic_dds_scaled through bhw_sincos_device vs the restatement of the two entities."""
    import blackman_harris_win_amd as bhw
    p = B.make_params(1, pw, w, model=model)
    n = 1 << pw
    for t0, cnt in ((0, min(n, 3000)), (n // 4 - 700, 1400), (n // 2 - 5, 1500), (3 * (n // 4) - 33, 900), (n - 1000, 2100)):
        t0 = max(t0, 0)
        s, c = bhw.cordic(p, t0, cnt)
        ws, wc = O.sincos(O.from_bhw(p), t0, cnt)
        assert np.array_equal(s.cpu().numpy(), ws) and np.array_equal(c.cpu().numpy(), wc), (t0, cnt)
    with pytest.raises(B.BhwError):                                    # sin/cos sources only
        gpu_generate(p, 0, 16)


@pytest.mark.parametrize("P,IW,AW", [(1, 23, 24), (3, 16, 16), (4, 32, 32), (2, 15, 16), (7, 25, 12), (1, 3, 4), (5, 31, 32)])
def test_atan2_matches_oracle(torch, P, IW, AW):
    import blackman_harris_win_amd as bhw
    rng = np.random.default_rng(P * 1000 + AW)
    lo, hi = -(1 << (IW - 1)), (1 << (IW - 1)) - 1
    x = rng.integers(lo, hi + 1, 6000)
    y = rng.integers(lo, hi + 1, 6000)
    edge = np.array([0, 1, -1, lo, hi, lo + 1, hi - 1, 1 << max(AW - 2, 0), -(1 << max(AW - 2, 0))], dtype=np.int64)
    edge = edge[(edge >= lo) & (edge <= hi)]
    x = np.concatenate([x, np.repeat(edge, len(edge))])
    y = np.concatenate([y, np.tile(edge, len(edge))])
    tx = torch.tensor(x.astype(np.int32), device="cuda")
    ty = torch.tensor(y.astype(np.int32), device="cuda")
    got = bhw.atan2(tx, ty, PRECISION=P, INPUT_WIDTH=IW, ANGLE_WIDTH=AW).cpu().numpy()
    assert np.array_equal(got, O.atan2(P, IW, AW, x, y))


def test_atan2_rejects_unbuildable_generics(torch):
    import blackman_harris_win_amd as bhw
    z = torch.zeros(4, dtype=torch.int32, device="cuda")
    with pytest.raises(B.BhwError) as e:
        bhw.atan2(z, z, PRECISION=1, INPUT_WIDTH=20, ANGLE_WIDTH=24)     # upstream defaults: VEC_DX(22) does not exist
    assert e.value.code == -2
    with pytest.raises(B.BhwError):
        bhw.atan2(z, z, PRECISION=0, INPUT_WIDTH=20, ANGLE_WIDTH=16)


def test_golden_reference_sincos(torch, golden, golden_dir):
    """GPU cordic() vs vectors produced by the reference's own compiled cordic() (model CPP)."""
    import blackman_harris_win_amd as bhw
    sc = np.load(os.path.join(golden_dir, golden["coe_cpp_14_12"]["file"])).astype(np.int32)
    s, c = bhw.cordic(B.make_params(1, 14, 12, model=B.MODEL_CPP), 0, 1 << 14)
    assert np.array_equal(s.cpu().numpy(), sc[:, 0]) and np.array_equal(c.cpu().numpy(), sc[:, 1])
    for name in [k for k in golden if k.startswith("sincos_cpp_")]:
        th, gs, gc = np.load(os.path.join(golden_dir, golden[name]["file"]))
        pr = golden[name]["params"]
        p = B.make_params(1, pr["phi_width"], pr["dat_width"], model=B.MODEL_CPP)
        s, c = bhw.cordic(p, 0, 1 << pr["phi_width"]) if pr["phi_width"] <= 20 else (None, None)
        if s is not None:
            s, c = s.cpu().numpy(), c.cpu().numpy()
            assert np.array_equal(s[th], gs) and np.array_equal(c[th], gc), name
        else:  # 2^24 / 2^26 phases: sweep in slices around the requested phases
            for t, es, ec in zip(th[::7], gs[::7], gc[::7]):
                s1, c1 = bhw.cordic(p, int(t), 1)
                assert (int(s1[0]), int(c1[0])) == (int(es), int(ec)), (name, int(t))


# ---- oracle comparisons on swept parameter sets -----------------------------------------------------------
CASES = []
for model in (B.MODEL_HLS, B.MODEL_CPP, B.MODEL_VHDL):
    for combine in (B.COMBINE_HLS, B.COMBINE_VHDL):
        for win, pw, w in [(1, 10, 16), (2, 10, 24), (3, 9, 12), (4, 12, 24), (5, 11, 30), (7, 12, 32), (7, 13, 31),
                           (4, 8, 8), (7, 10, 18), (5, 14, 13), (7, 4, 16), (4, 18, 17)]:
            if model == B.MODEL_HLS and pw > w + 2:
                continue
            CASES.append((model, combine, win, pw, w))


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("model,combine,win,pw,w", CASES)
def test_window_matches_oracle(torch, model, combine, win, pw, w, algo):
    prec = 1 + (pw + w) % 3 if model == B.MODEL_VHDL else 1
    p = B.make_params(win, pw, w, model=model, combine=combine, precision=prec)
    n = 1 << pw
    count = min(n, 4096)
    n0 = 0 if n <= 4096 else (n // 2 - 1000)
    want = O.generate(O.from_bhw(p), n0, count)
    got = gpu_generate(p, n0, count, algo)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("model", [B.MODEL_HLS, B.MODEL_CPP, B.MODEL_VHDL])
@pytest.mark.parametrize("pw,w", [(10, 16), (12, 12), (14, 32), (16, 30), (13, 31), (20, 24), (26, 32), (10, 8)])
def test_sincos_matches_oracle(torch, model, pw, w):
    import blackman_harris_win_amd as bhw
    p = B.make_params(1, pw, w, model=model, precision=2 if model == B.MODEL_VHDL else 1)
    n = 1 << pw
    rng = np.random.default_rng(pw * 100 + w)
    starts = [0, n // 4 - 50, n // 2 - 50, 3 * (n // 4) - 50, n - 100] + [int(v) for v in rng.integers(0, n, 3)]
    for t0 in starts:
        cnt = min(100, n)
        s, c = bhw.cordic(p, t0, cnt)
        ws, wc = O.sincos(O.from_bhw(p), t0, cnt)
        assert np.array_equal(s.cpu().numpy(), ws) and np.array_equal(c.cpu().numpy(), wc), (model, pw, w, t0)


def test_custom_weights_and_wrapping_sum(torch):
    """AA ports are caller-scaled: weights large enough to overflow W bits must wrap exactly like win_t."""
    aa = [(1 << 22) + 12345, (1 << 22) - 1, 1 << 20, 777777]
    for combine in (B.COMBINE_HLS, B.COMBINE_VHDL):
        for algo in ALGOS:
            p = B.make_params(4, 11, 24, combine=combine, aa=aa)
            assert np.array_equal(gpu_generate(p, 0, 2048, algo), O.generate(O.from_bhw(p), 0, 2048))
    p = B.make_params(7, 10, 32, aa=[2**31 - 1, -(2**31), 2**31 - 1, -12345, 2**30, -(2**30), 7])
    for algo in ALGOS:
        assert np.array_equal(gpu_generate(p, 0, 1024, algo), O.generate(O.from_bhw(p), 0, 1024))


# ---- whole-period calls: fold kernel below 2^22 coefficients, gather tiles (1-, 3-, 15-run; 32/64-bit sums) from 2^22 ----
TILE_CASES = [(1, 16, 16, B.MODEL_HLS, B.COMBINE_HLS), (3, 16, 24, B.MODEL_HLS, B.COMBINE_VHDL),
              (4, 17, 24, B.MODEL_CPP, B.COMBINE_HLS), (5, 16, 32, B.MODEL_HLS, B.COMBINE_HLS),
              (7, 16, 32, B.MODEL_VHDL, B.COMBINE_VHDL), (7, 17, 30, B.MODEL_CPP, B.COMBINE_VHDL),
              (7, 16, 31, B.MODEL_HLS, B.COMBINE_VHDL), (4, 18, 16, B.MODEL_HLS, B.COMBINE_HLS),
              (7, 18, 20, B.MODEL_VHDL, B.COMBINE_HLS), (2, 16, 24, B.MODEL_HLS, B.COMBINE_HLS),
              # phase bits dropped (PW >= W): small shared table, one-run form of the tile kernel
              (7, 20, 12, B.MODEL_CPP, B.COMBINE_HLS), (7, 18, 16, B.MODEL_VHDL, B.COMBINE_VHDL),
              (5, 19, 14, B.MODEL_CPP, B.COMBINE_VHDL), (4, 18, 16, B.MODEL_HLS, B.COMBINE_HLS),
              # 2^20, 2^21: still the fold kernel (the 960-thread tiles only win from 2^22 on)
              (1, 20, 16, B.MODEL_CPP, B.COMBINE_HLS), (7, 21, 12, B.MODEL_CPP, B.COMBINE_VHDL),
              # N >= 2^22: tile kernel proper (1-, 3-, 15-run tiles; 32- and 64-bit sums; plain and packed tables; dropped phase bits)
              (1, 22, 16, B.MODEL_CPP, B.COMBINE_HLS), (3, 22, 24, B.MODEL_HLS, B.COMBINE_VHDL),
              (5, 22, 30, B.MODEL_VHDL, B.COMBINE_VHDL), (7, 22, 12, B.MODEL_CPP, B.COMBINE_VHDL),
              (7, 22, 30, B.MODEL_HLS, B.COMBINE_HLS), (7, 23, 32, B.MODEL_CPP, B.COMBINE_HLS),
              (7, 22, 32, B.MODEL_VHDL, B.COMBINE_HLS), (4, 22, 32, B.MODEL_HLS, B.COMBINE_HLS),
              # VHDL cosine-sum in the 15-run tiles (W+2-bit sum carried as 4*hi + lo), both quadrant maps, plain and packed tables
              (7, 22, 30, B.MODEL_HLS, B.COMBINE_VHDL), (7, 22, 28, B.MODEL_CPP, B.COMBINE_VHDL),
              (7, 22, 32, B.MODEL_VHDL, B.COMBINE_VHDL), (2, 22, 31, B.MODEL_HLS, B.COMBINE_VHDL),
              # fewest rotations a packed table can meet: VHDL model at PW == W (z_shr = 0, W - 1 = 21 rotations; found by the fuzzer)
              (2, 22, 22, B.MODEL_VHDL, B.COMBINE_VHDL), (7, 22, 22, B.MODEL_HLS, B.COMBINE_HLS)]


@pytest.mark.parametrize("win,pw,w,model,combine", TILE_CASES)
def test_whole_period_tile_path(torch, win, pw, w, model, combine):
    p = B.make_params(win, pw, w, model=model, combine=combine)
    n = 1 << pw
    want = O.generate_mt(O.from_bhw(p), 0, n)
    assert np.array_equal(gpu_generate(p, 0, n, B.ALGO_TABLE), want)
    # two periods: the second is the store-only replica of the first
    two = gpu_generate(p, n, 2 * n, B.ALGO_TABLE)
    assert np.array_equal(two[:n], want) and np.array_equal(two[n:], want)


@pytest.mark.parametrize("combine", [B.COMBINE_HLS, B.COMBINE_VHDL])
@pytest.mark.parametrize("win,w", [(7, 30), (5, 24), (2, 32)])
def test_tile_path_with_wrapping_caller_weights(torch, win, w, combine):
    """AA0..AA6 are ports: full-range random weights make the W / W+1 / W+2-bit sums of both rules wrap; tile kernels at 2^22."""
    rng = np.random.default_rng(win * 100 + w + combine)
    aa = [int(v) for v in rng.integers(-(1 << (w - 1)), 1 << (w - 1), 7)]
    p = B.make_params(win, 22, w, combine=combine, aa=aa)
    n = 1 << 22
    assert np.array_equal(gpu_generate(p, 0, n, B.ALGO_TABLE), O.generate_mt(O.from_bhw(p), 0, n))


@pytest.mark.gpu
@pytest.mark.parametrize("w", [32, 28])
def test_weights_at_the_edge_of_the_one_instruction_products(torch, w):
    """The tile and fused kernels multiply by the pre-shifted weight a << (34 - W) and by its negation when every |a_k| is below
    2^(W-3): weights at +-(2^(W-3) - 1) take that form, a weight of exactly -2^(W-3) (whose negation does not fit) must not."""
    import blackman_harris_win_amd as bhw
    lim = 1 << (w - 3)
    n = 1 << 22
    for aa in ([lim - 1, -(lim - 1), lim - 1, -(lim - 1), lim - 1, -(lim - 1), lim - 1],
               [lim - 1, -lim, lim - 1, 12345, -lim, 777, -lim]):
        p = B.make_params(7, 22, w, aa=aa)
        want = O.generate_mt(O.from_bhw(p), 0, n)
        assert np.array_equal(gpu_generate(p, 0, n, B.ALGO_TABLE), want), aa
        assert np.array_equal(bhw.generate(p, 0, n, algo=B.ALGO_FUSED).cpu().numpy(), want), aa
        # the VHDL cosine-sum takes the same one-instruction products in the tile kernel ((q + 1) >> 1 on q = mul_hi), with either
        # quadrant negation (model cpp: ~v)
        for model in (B.MODEL_HLS, B.MODEL_CPP):
            pv = B.make_params(7, 22, w, aa=aa, combine=B.COMBINE_VHDL, model=model)
            assert np.array_equal(gpu_generate(pv, 0, n, B.ALGO_TABLE), O.generate_mt(O.from_bhw(pv), 0, n)), (aa, model)


@pytest.mark.gpu
def test_vhdl_rule_one_word_sums_at_the_bound(torch):
    """VHDL cosine-sum in the tile kernel: with one-instruction products the W+2-bit sum is kept in ONE 32-bit word when the sum
    of the (|a_k| + 1) stays below 2^31, in two words otherwise -- weights on either side of that bound, both signs of a_0."""
    h = 1 << 28
    n = 1 << 22
    for a0 in ((1 << 29) - 8, (1 << 29) - 7, -((1 << 29) - 8), -((1 << 29) - 7)):
        for harm in ([h, -h, h, -h, h, -h], [-h, -h, -h, -h, -h, -h], [h, h, h, h, h, h]):
            for model in (B.MODEL_HLS, B.MODEL_CPP):
                pv = B.make_params(7, 22, 32, aa=[a0] + harm, combine=B.COMBINE_VHDL, model=model)
                assert np.array_equal(gpu_generate(pv, 0, n, B.ALGO_TABLE), O.generate_mt(O.from_bhw(pv), 0, n)), (a0, harm, model)


def test_strategies_agree_on_random_whole_windows(torch):
    """Size-independent property: DIRECT (one CORDIC chain per harmonic per coefficient) and TABLE (shared table,
    folds, gather tiles) are different computations of the same integers -- whole windows must be identical."""
    import blackman_harris_win_amd as bhw
    rng = np.random.default_rng(2024)
    for _ in range(24):
        win = int(rng.choice([1, 2, 3, 4, 5, 7]))
        model = int(rng.integers(0, 3))
        combine = int(rng.integers(0, 2))
        w = int(rng.integers(8, 33))
        pw = int(rng.integers(14, 23))
        if model == B.MODEL_HLS and pw > w + 2:
            pw = w + 2
        prec = int(rng.integers(1, 3)) if model == B.MODEL_VHDL else 1
        aa = [int(v) for v in rng.integers(-(1 << (w - 1)), 1 << (w - 1), 7)] if rng.random() < 0.5 else None
        p = B.make_params(win, pw, w, model=model, combine=combine, precision=prec, aa=aa)
        a = bhw.generate(p, 0, 1 << pw, algo=B.ALGO_DIRECT)
        b = bhw.generate(p, 0, 1 << pw, algo=B.ALGO_TABLE)
        assert bool((a == b).all()), (win, model, combine, w, pw, prec, aa)
        # and an unaligned slice through the general (non-fold) table path
        n0, cnt = int(rng.integers(1, 1 << pw)), int(rng.integers(1, 5000))
        c = bhw.generate(p, n0, cnt, algo=B.ALGO_TABLE)
        ref = torch.cat([a, a, a])[n0:n0 + cnt] if n0 + cnt <= 3 << pw else None
        assert ref is None or bool((c == ref).all()), (win, model, combine, w, pw, n0, cnt)


# ---- edge cases: empty, ragged, offsets, wrap-around --------------------------------------------------------
@pytest.mark.parametrize("algo", ALGOS)
def test_ragged_counts_and_offsets(torch, algo):
    p = B.make_params(7, 12, 32)
    po = O.from_bhw(p)
    n = 1 << 12
    for n0, count in [(0, 1), (1, 1), (5, 255), (7, 257), (n - 3, 10), (3 * n + 17, 513), (2**40 + 5, 64), (123, 1000)]:
        assert np.array_equal(gpu_generate(p, n0, count, algo), O.generate(po, n0, count)), (n0, count)


@pytest.mark.parametrize("win,pw,w,model,n0,count", [
    (7, 12, 32, B.MODEL_HLS, 1000, 3 * 4096 + 123),            # head + 2 periods (fold + replicate) + tail
    (4, 10, 24, B.MODEL_CPP, 1023, 1025),                       # 1-sample head, one period, no tail
    (5, 11, 16, B.MODEL_VHDL, 2048 * 7 + 1, 2047 + 2048),       # head + exactly one period
    (7, 20, 32, B.MODEL_HLS, 12345, (1 << 20) + (1 << 19)),     # fold kernel in the middle of a ragged range
    (7, 22, 30, B.MODEL_HLS, 54321, (1 << 22) + (1 << 20)),     # tile kernel (packed table) in the middle of a ragged range
])
def test_ragged_range_spanning_whole_periods(torch, win, pw, w, model, n0, count):
    p = B.make_params(win, pw, w, model=model)
    want = O.generate_mt(O.from_bhw(p), n0, count)
    assert np.array_equal(gpu_generate(p, n0, count, B.ALGO_TABLE), want)


def test_empty_and_null_arguments(torch):
    L = B.lib()
    p = B.make_params(4, 12, 24)
    assert L.bhw_generate_device(ctypes.byref(p), 0, None, 0, 0, None) == 0          # count 0: nothing to do
    assert L.bhw_generate_device(ctypes.byref(p), 0, None, 0, 16, None) == -1        # NULL output
    assert L.bhw_sincos_device(ctypes.byref(p), 0, None, 0, 16, None, None) == -1
    assert L.bhw_generate_device(ctypes.byref(p), 99, None, 0, 16, ctypes.c_void_p(16)) == -3   # no such device
    host = (ctypes.c_int32 * 100)()
    assert L.bhw_generate_to_host(ctypes.byref(p), 0, 5, 100, host) == 0
    assert list(host) == list(O.generate(O.from_bhw(p), 5, 100))


def test_explicit_workspace_and_stream(torch):
    import blackman_harris_win_amd as bhw
    p = B.make_params(7, 16, 32)
    need = B.lib().bhw_workspace_bytes(ctypes.byref(p), 0, 1 << 16, B.ALGO_TABLE)
    assert need == (1 << 14) * 8
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    want = O.generate(O.from_bhw(p), 0, 1 << 16)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        got = bhw.generate(p, 0, 1 << 16, algo=B.ALGO_TABLE, workspace=ws)
    st.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)
    small = torch.empty(16, dtype=torch.uint8, device="cuda")
    with pytest.raises(B.BhwError) as ei:
        bhw.generate(p, 0, 1 << 16, algo=B.ALGO_TABLE, workspace=small)
    assert ei.value.code == -4


@pytest.mark.parametrize("win,pw,w", [(4, 20, 24), (7, 22, 30)])
def test_hip_graph_capture_with_caller_workspace(torch, win, pw, w):
    """With a caller-owned workspace the launch path allocates nothing and never synchronises, so a call can be captured into
    a HIP graph and replayed (fold path at 2^20, packed-table tile path at 2^22)."""
    import blackman_harris_win_amd as bhw
    p = B.make_params(win, pw, w)
    n = 1 << pw
    want = bhw.generate(p, 0, n).clone()                                   # also warms every lazy initialisation
    ws = torch.empty(B.lib().bhw_workspace_bytes(ctypes.byref(p), 0, n, B.ALGO_AUTO), dtype=torch.uint8, device="cuda")
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        bhw.generate(p, 0, n, out=out, workspace=ws)
    for _ in range(3):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert bool((out == want).all())


def test_concurrent_streams_use_separate_library_scratch(torch):
    """Two different windows generated concurrently on two streams with library-owned scratch must not share a table."""
    import blackman_harris_win_amd as bhw
    pa, pb = B.make_params(7, 18, 32), B.make_params(5, 18, 24, model=B.MODEL_CPP)
    wa, wb = O.generate(O.from_bhw(pa), 0, 1 << 18), O.generate(O.from_bhw(pb), 0, 1 << 18)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(4):
        with torch.cuda.stream(sa):
            ga = bhw.generate(pa, 0, 1 << 18, algo=B.ALGO_TABLE)
        with torch.cuda.stream(sb):
            gb = bhw.generate(pb, 0, 1 << 18, algo=B.ALGO_TABLE)
        torch.cuda.synchronize()
        assert np.array_equal(ga.cpu().numpy(), wa) and np.array_equal(gb.cpu().numpy(), wb)
    assert B.lib().bhw_release_device(0) == 0


# ---- fused apply y = (x * w) >> shift ------------------------------------------------
@pytest.mark.parametrize("win,pw,w,model,combine,n0,count,shift", [
    (7, 16, 32, B.MODEL_HLS, B.COMBINE_HLS, 0, 1 << 16, 31),        # whole period, tile path
    (7, 16, 32, B.MODEL_HLS, B.COMBINE_HLS, 0, 3 << 16, 30),        # three periods of samples, one table
    (4, 17, 24, B.MODEL_CPP, B.COMBINE_HLS, 12345, 50000, 23),      # unaligned range, general table path
    (5, 12, 16, B.MODEL_VHDL, B.COMBINE_VHDL, 0, 4096, 15),         # whole period, plain fold path
    (3, 10, 24, B.MODEL_HLS, B.COMBINE_HLS, 7, 700, 0),             # short: direct path, shift 0
    (1, 12, 16, B.MODEL_HLS, B.COMBINE_VHDL, 0, 4096, 62),          # extreme shift
])
def test_fused_apply_matches_oracle(torch, win, pw, w, model, combine, n0, count, shift):
    import blackman_harris_win_amd as bhw
    p = B.make_params(win, pw, w, model=model, combine=combine)
    rng = np.random.default_rng(count + shift)
    x = rng.integers(-(1 << 31), 1 << 31, count, dtype=np.int64).astype(np.int32)
    wv = O.generate(O.from_bhw(p), n0, count).astype(np.int64)
    want = ((x.astype(np.int64) * wv) >> shift).astype(np.int32)            # exact product, floor shift, low 32 bits
    got = bhw.apply(p, torch.from_numpy(x).cuda(), n0=n0, shift=shift).cpu().numpy()
    assert np.array_equal(got, want)


def test_fused_apply_full_window_tile_path(torch):
    """The fused multiplier stage on the tile kernels (2^22, packed table): y = (x * w) >> shift with the exact 64-bit product."""
    import blackman_harris_win_amd as bhw
    p = B.make_params(7, 22, 30)
    n = 1 << 22
    x = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda")
    y = bhw.apply(p, x, shift=29)
    w = bhw.generate(p, 0, n)
    assert bool((((x.to(torch.int64) * w.to(torch.int64)) >> 29).to(torch.int32) == y).all())


def test_fused_apply_rejects_aliasing(torch):
    p = B.make_params(4, 12, 24)
    x = torch.zeros(4096, dtype=torch.int32, device="cuda")
    L = B.lib()
    rc = L.bhw_apply_device(ctypes.byref(p), 0, None, 0, 4096, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(x.data_ptr()), 23)
    assert rc == -1 and b"overlap" in L.bhw_last_error()
    assert L.bhw_apply_device(ctypes.byref(p), 0, None, 0, 4096, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(x.data_ptr() + 4 * 4096), 63) == -1


# ---- the selector / HLS-top mirrors -------------------------------------------------------------------
def test_win_selector_streaming_counter(torch):
    from blackman_harris_win_amd import WinSelector
    sel = WinSelector(PHI_WIDTH=10, DAT_WIDTH=24, WIN_TYPE="BH5TERM")
    full = O.generate(O.oparams(5, 10, 24), 0, 1024)
    a = sel.enable(700).cpu().numpy()
    b = sel.enable(700).cpu().numpy()        # wraps past the end of the period
    assert np.array_equal(a, full[:700])
    assert np.array_equal(b, np.concatenate([full[700:], full[:376]]))
    sel.reset()
    assert np.array_equal(sel.window().cpu().numpy(), full)
    x = torch.arange(-500, 524, dtype=torch.int32, device="cuda") * 1000003
    sel.reset()
    sel.enable(100)
    y = sel.apply(x, shift=23).cpu().numpy()                   # continues from phase 100 and wraps
    wv = np.concatenate([full[100:], full[:100]]).astype(np.int64)
    assert np.array_equal(y, ((x.cpu().numpy().astype(np.int64) * wv) >> 23).astype(np.int32))
    sel.reset()
    s0 = sel.shard(0, 2).cpu().numpy()
    s1 = sel.shard(1, 2).cpu().numpy()
    assert np.array_equal(np.concatenate([s0, s1]), full)


def test_win_function_unknown_type_is_zero(torch):
    from blackman_harris_win_amd import win_function
    assert int(win_function(6, 0, 64, nphase=10, nwidth=16).abs().sum()) == 0     # win_empty
    got = win_function(5, 0, 1024, nphase=10, nwidth=24).cpu().numpy()
    assert _md5(got) == "46e784b28a4e1a74fb90e8cc13978b4c"


def test_batched_frames(torch, golden):
    import blackman_harris_win_amd as bhw
    p = B.make_params(4, 16, 24)
    out = bhw.generate_batched(p, 37)
    frame = out[0].cpu().numpy()
    assert _md5(frame) == golden["C4_bh4_16_24_frame"]["md5"]
    assert bool((out == out[0:1]).all())
    # the replicate path equals recomputing the stream (n wraps mod 2^phi_width)
    again = bhw.generate(p, 0, 5 << 16, algo=B.ALGO_DIRECT).view(5, -1)
    assert bool((again == out[:5]).all())
    tiny = bhw.generate_batched(B.make_params(7, 4, 16), 3).cpu().numpy()
    assert np.array_equal(tiny, np.tile(O.generate(O.oparams(7, 4, 16), 0, 16), (3, 1)))


@pytest.mark.parametrize("cfg", [
    # (win, pw, w, model, combine, frames): the one-launch form (periods up to 2^19 the fused kernel takes: every frame written by
    # the kernel that computes the period) with frame counts that do and do not divide over its workgroup rows ...
    (4, 16, 24, B.MODEL_HLS, B.COMBINE_HLS, 37), (4, 14, 24, B.MODEL_CPP, B.COMBINE_VHDL, 513), (7, 12, 16, B.MODEL_VHDL, B.COMBINE_VHDL, 1000),
    (5, 18, 28, B.MODEL_HLS, B.COMBINE_HLS, 9), (3, 19, 20, B.MODEL_CPP, B.COMBINE_HLS, 5), (2, 10, 12, B.MODEL_HLS, B.COMBINE_VHDL, 2),
    # ... and periods that keep one period + replicate: 2^20, the chains-split form (BH-7 2^16 at 32 bits), a 34-bit-plus state
    (4, 20, 24, B.MODEL_HLS, B.COMBINE_HLS, 3), (7, 16, 32, B.MODEL_HLS, B.COMBINE_HLS, 11), (7, 12, 32, B.MODEL_VHDL, B.COMBINE_HLS, 6)])
def test_batched_frames_both_forms_match_the_oracle(torch, cfg):
    """bhw_generate_batched_device: `frames` identical periods (the stream is periodic, src/bh_win_7term.vhd:92-97), bit-exact in
    every frame whichever way they are produced."""
    import blackman_harris_win_amd as bhw
    win, pw, w, model, combine, frames = cfg
    p = B.make_params(win, pw, w, model=model, combine=combine, precision=3 if (model == B.MODEL_VHDL and w == 32) else 1)
    out = bhw.generate_batched(p, frames)
    assert out.shape == (frames, 1 << pw)
    want = O.generate_mt(O.from_bhw(p), 0, 1 << pw)
    assert np.array_equal(out[0].cpu().numpy(), want) and np.array_equal(out[frames - 1].cpu().numpy(), want)
    assert bool((out == out[0:1]).all())


# ---- BASELINE full sizes: checksums + size-independent properties -----------------------------------
def test_c2_full(torch, golden):
    p = B.make_params(4, 20, 24)
    for algo in ALGOS:
        got = gpu_generate(p, 0, 1 << 20, algo)
        assert _md5(got) == golden["C2_bh4_20_24"]["md5"]


def test_c4_full_1024_frames(torch, golden):
    import blackman_harris_win_amd as bhw
    out = bhw.generate_batched(B.make_params(4, 16, 24), 1024)
    assert out.shape == (1024, 65536)
    assert _md5(out[1023].cpu().numpy()) == golden["C4_bh4_16_24_frame"]["md5"]
    assert bool((out == out[0:1]).all())
    assert int(out.sum(dtype=torch.int64)) == 1024 * golden["C4_bh4_16_24_frame"]["sum"]


def test_c3_full_64m(torch, golden):
    """BH-7, N = 2^26, 32-bit: whole window through the table strategy, per-shard md5 vs golden."""
    import blackman_harris_win_amd as bhw
    e = golden["C3_bh7_26_32"]
    p = B.make_params(7, 26, 32)
    full = bhw.generate(p, 0, 1 << 26, algo=B.ALGO_TABLE)
    assert int(full.min()) == 65 and int(full.max()) == 1073741825
    for g in range(8):
        sh = full[g << 23:(g + 1) << 23]
        assert int(sh.sum(dtype=torch.int64)) == e["shards"][g]["sum"]
        assert _md5(sh.cpu().numpy()) == e["shards"][g]["md5"]
    for n, v in e["sparse"].items():
        assert int(full[int(n)]) == v
    # C5 sharding: each device-sized shard generated on its own (both strategies) equals the slice of the whole
    for g in range(8):
        for algo in ALGOS + [B.ALGO_AUTO]:
            sh = bhw.generate(p, g << 23, 1 << 23, algo=algo)
            assert bool((sh == full[g << 23:(g + 1) << 23]).all()), (g, algo)
            assert _md5(sh.cpu().numpy()) == e["shards"][g]["md5"]
    # periodicity: the stream index wraps modulo N
    tail = bhw.generate(p, (1 << 26) - 1000, 2000, algo=B.ALGO_DIRECT)
    assert bool((tail[:1000] == full[-1000:]).all()) and bool((tail[1000:] == full[:1000]).all())
    del full


def test_packed_table_is_exact(torch, golden):
    """The packed tables -- "nibble" / "nibble + escapes" / "residual" (1 / 1 / 2 bytes per entry against a linear predictor) and
    "delta16" (int16 differences to the first entry of each 64-entry block) -- must give the same coefficients as the plain int2
    table (bhw_exec.table_format)."""
    import blackman_harris_win_amd as bhw
    for win, pw, w, model in ((7, 26, 32, 0), (7, 24, 32, 1), (4, 22, 24, 2), (5, 23, 29, 0), (7, 22, 28, 2), (3, 22, 30, 0),
                              (7, 22, 32, 0), (7, 25, 26, 1), (7, 26, 32, 1), (7, 26, 32, 2), (5, 25, 31, 1), (7, 24, 24, 2)):
        p = B.make_params(win, pw, w, model=model)
        # a ragged head, one whole period (the tile kernel over the packed table) and a ragged tail (the general gather over the same table)
        outs = [bhw.generate(p, (1 << pw) - 12345, (1 << pw) + 12345 + 999, algo=B.ALGO_TABLE, table_format=f)
                for f in (B.TABLE_PLAIN, B.TABLE_DELTA16, B.TABLE_RESIDUAL, B.TABLE_NIBBLE_ESC, B.TABLE_NIBBLE, B.TABLE_BEST)]
        for o in outs[1:]:
            assert bool((o == outs[0]).all()), (win, pw, w, model)
        del outs
    # the VHDL cosine-sum over the same formats (k_tile9's one-instruction products at 32 bits, plain nibbles and nibble + escapes -- the
    # VHDL CORDIC's table lists ~900 entries: the end-of-harmonic repair of marked lanes runs in a few per cent of the waves -- and
    # the general rounding below 32 bits)
    for win, pw, w, model in ((7, 26, 32, 2), (7, 26, 32, 0), (7, 26, 32, 1), (7, 26, 30, 2)):
        p = B.make_params(win, pw, w, model=model, combine=B.COMBINE_VHDL)
        outs = [bhw.generate(p, 0, 1 << pw, algo=B.ALGO_TABLE, table_format=f) for f in (B.TABLE_PLAIN, B.TABLE_NIBBLE_ESC, B.TABLE_BEST)]
        for o in outs[1:]:
            assert bool((o == outs[0]).all()), (win, pw, w, model, "vhdl sum")
        for a in (0, (1 << pw) // 2 - 2048, (1 << pw) - 4096):
            assert np.array_equal(outs[0][a:a + 4096].cpu().numpy(), O.generate_mt(O.from_bhw(p), a, 4096)), (win, pw, w, model, a)
        del outs


def test_nibble_escape_tables_in_every_tile_path(torch):
    """The cpp model at 2^24 / 32 bits: CORDIC noise a little wider than the nibble fields, so BEST settles on nibble + escapes (one
    byte per entry, a marker for the deviation that does not fit, the exact pair in the build workgroup's hash table).  Every kernel
    that reads the table must resolve the marker: the 15-run tiles of a whole window, the masked instance of an image subset, the
    part instances, the fused apply and the general gather of ragged ends -- all against the plain table, itself against the oracle."""
    import blackman_harris_win_amd as bhw
    p = B.make_params(7, 24, 32, model=B.MODEL_CPP)
    n = 1 << 24
    plain = bhw.generate(p, 0, n, algo=B.ALGO_TABLE, table_format=B.TABLE_PLAIN)
    for a in (0, n // 2 - 4096, n - 8192):
        assert np.array_equal(plain[a:a + 8192].cpu().numpy(), O.generate_mt(O.from_bhw(p), a, 8192))
    full = bhw.generate(p, 0, n, algo=B.ALGO_TABLE)
    assert B.describe_plan(p, 0, n, algo=B.ALGO_TABLE).startswith("table[nibble+esc]"), B.describe_plan(p, 0, n, algo=B.ALGO_TABLE)
    assert bool((full == plain).all())
    for n0, count in ((n // 8, n // 4), (5 * (n // 8), 3 * (n // 8)), (n - 777, n + 777 + 55)):
        sh = bhw.generate(p, n0, count, algo=B.ALGO_TABLE, table_format=B.TABLE_NIBBLE_ESC)
        idx = (torch.arange(count, device="cuda") + n0) % n
        assert bool((sh == plain[idx]).all()), (n0, count)
    window = torch.zeros(n, dtype=torch.int32, device="cuda")
    for part in range(3):
        bhw.generate_part(p, part, 3, window, algo=B.ALGO_TABLE, table_format=B.TABLE_NIBBLE_ESC)
    assert bool((window == plain).all())
    x = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int32, device="cuda")
    y = bhw.apply(p, x, shift=31)
    assert bool((((x.to(torch.int64) * plain.to(torch.int64)) >> 31).to(torch.int32) == y).all())


def test_c3_full_64m_model_cpp(torch, golden):
    """The headline window with the cpp model's cosines (the model pinned by the reference's own cordic() build)."""
    import blackman_harris_win_amd as bhw
    e = golden["C3cpp_bh7_26_32"]
    p = B.make_params(7, 26, 32, model=B.MODEL_CPP)
    full = bhw.generate(p, 0, 1 << 26, algo=B.ALGO_TABLE)
    for g in range(8):
        sh = full[g << 23:(g + 1) << 23]
        assert int(sh.sum(dtype=torch.int64)) == e["shards"][g]["sum"]
        assert _md5(sh.cpu().numpy()) == e["shards"][g]["md5"]
    del full


def test_c3_model_cpp_full_sincos_quadrant_property(torch):
    """2^26 phases at 32 bits, model CPP: the three upper quadrants are the quadrant-mapped first one."""
    import blackman_harris_win_amd as bhw
    p = B.make_params(1, 26, 32, model=B.MODEL_CPP)
    q = 1 << 24
    s0, c0 = bhw.cordic(p, 0, q)
    s1, c1 = bhw.cordic(p, q, q)
    assert bool((c1 == ~s0).all()) and bool((s1 == c0).all())
    s3, c3 = bhw.cordic(p, 3 * q, q)
    assert bool((c3 == s0).all()) and bool((s3 == ~c0).all())
    # spot check against the oracle (pinned to the reference's cordic())
    ws, wc = O.sincos(O.from_bhw(p), q - 500, 1000)
    s, c = bhw.cordic(p, q - 500, 1000)
    assert np.array_equal(s.cpu().numpy(), ws) and np.array_equal(c.cpu().numpy(), wc)


@pytest.mark.parametrize("model", [B.MODEL_HLS, B.MODEL_CPP, B.MODEL_VHDL])
def test_whole_period_sincos_sweep_through_shared_prefixes(torch, model):
    """cordic() over exactly one period from 2^16 phases on runs the shared-prefix chains of the table build with the four
    quadrant images written straight out (bhwk_sincos): every phase against the oracle, any start phase, either output alone;
    the per-phase kernel (count != N) gives the same values."""
    import blackman_harris_win_amd as bhw
    for pw, w, theta0 in ((16, 16, 0), (18, 24, 12345), (22, 32, (1 << 22) - 7), (20, 18, 3 << 20)):
        if model == B.MODEL_HLS and pw > w + 2:
            continue
        p = B.make_params(1, pw, w, model=model, precision=2 if model == B.MODEL_VHDL else 1)
        n = 1 << pw
        ws, wc = O.sincos_mt(O.from_bhw(p), theta0, n)
        gs, gc = bhw.cordic(p, theta0, n)
        assert np.array_equal(gs.cpu().numpy(), ws) and np.array_equal(gc.cpu().numpy(), wc), (model, pw, w, theta0)
        hs, hc = bhw.cordic(p, theta0, n + 1)                       # one more phase: the per-phase kernel
        assert bool((hs[:n] == gs).all()) and bool((hc[:n] == gc).all())


@pytest.mark.parametrize("model", [B.MODEL_DDS48, B.MODEL_SCALED])
def test_whole_period_sweep_of_the_variant_generators(torch, model):
    """cordic_dds48 / cordic_dds_scaled over exactly one period (k_prerot_sweep: shared rotation prefixes per quadrant) against
    the oracle, any start phase; one phase more takes the per-phase kernel and must give the same values."""
    import blackman_harris_win_amd as bhw
    for pw, w, theta0 in ((16, 16, 0), (18, 24, 54321), (20, 32, (1 << 20) - 3), (17, 8, 1 << 15), (22, 29, 7)):
        p = B.make_params(1, pw, w, model=model)
        n = 1 << pw
        ws, wc = O.sincos_mt(O.from_bhw(p), theta0, n)
        gs, gc = bhw.cordic(p, theta0, n)
        assert np.array_equal(gs.cpu().numpy(), ws) and np.array_equal(gc.cpu().numpy(), wc), (model, pw, w, theta0)
        hs, hc = bhw.cordic(p, theta0, n + 1)
        assert bool((hs[:n] == gs).all()) and bool((hc[:n] == gc).all())
