#!/usr/bin/env python3
"""Generate the committed golden fixtures (tests/golden/golden.json + small .npy vectors).

Run in the build container (needs /root/reference for the 'reference' entries):
    make -C oracle && python tests/golden/make_golden.py

Provenance classes recorded per entry:
  reference : produced by the reference's own code run here -- oracle/_ref = cordic() of
              cpp/cordic_sincos.cpp compiled from its source, and its main() writing coe.dat.
  survey    : known answers recorded in SURVEY.md App. B (obtained during the survey from the
              unmodified HLS sources); this script only re-states them and checks that the oracle
              reproduces every one before writing them out.
  oracle    : produced by the CPU restatement alone (model C / VHDL combine / Taylor have no
              runnable reference here -> "parity unpinned"; large-vector checksums of pinned models).
Fixtures are data (inputs + expected outputs); no reference source text is stored.
"""
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import tempfile
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402


def md5(a):
    return hashlib.md5(np.ascontiguousarray(a, dtype="<i4").tobytes()).hexdigest()


def stats(a):
    a64 = a.astype(np.int64)
    return {"md5": md5(a), "fnv1a64": O.fnv(a), "sum": int(a64.sum()), "min": int(a64.min()), "max": int(a64.max()),
            "count": int(a.size)}


def pdict(p):
    d = {f: int(getattr(p, f)) for f in "model combine sin_type n_terms phi_width dat_width precision lut_size".split()}
    d["aa"] = [int(v) for v in p.aa]
    return d


def c3_shard(g, model=O.MODEL_HLS):
    cache = f"/tmp/c3_shard{g}.npy" if model == O.MODEL_HLS else f"/tmp/c3_m{model}_shard{g}.npy"
    if os.path.exists(cache):
        a = np.load(cache)
    else:
        a = O.generate(O.oparams(7, 26, 32, model=model), g << 23, 1 << 23)
        np.save(cache, a)
    st = stats(a)
    st["strided_1024"] = [int(v) for v in a[:: (1 << 23) // 1024][:1024]]
    return g, st


def c3_shard_cpp(g):
    """Shard g of the headline window with model A cosines, evaluated by the reference's own compiled cordic()
    (oracle/_ref/libref_cordic_26_32.so) in the cosine-sum of win_function.cpp:361-375 -- no oracle restatement of the CORDIC."""
    a = O.reference_window(O.oparams(7, 26, 32, model=O.MODEL_CPP), g << 23, 1 << 23, threads=1)
    st = stats(a)
    st["strided_1024"] = [int(v) for v in a[:: (1 << 23) // 1024][:1024]]
    return g, st


def main():
    out = {"_readme": "see make_golden.py; fnv1a64 = bytewise FNV-1a-64 over the little-endian int32 vector",
           "entries": {}}
    E = out["entries"]

    # ---- reference: coe.dat of cpp/cordic_sincos.cpp main() at its in-file widths (14, 12) -------------
    ref = {(pw, w): path for pw, w, path in O.ref_pairs()}
    if (14, 12) in ref:
        with tempfile.TemporaryDirectory() as td:
            code = ("import ctypes,sys; l=ctypes.CDLL(%r); l._Z8ref_mainiPPc.argtypes=[ctypes.c_int,ctypes.c_void_p];"
                    "l._Z8ref_mainiPPc(0,None)" % ref[(14, 12)])
            subprocess.run([sys.executable, "-c", code], cwd=td, stdout=subprocess.DEVNULL, check=True)
            raw = open(os.path.join(td, "..\\math\\coe.dat"), "rb").read()
        sc = np.array([[int(v) for v in ln.split()] for ln in raw.decode().splitlines()], dtype=np.int32)
        assert sc.shape == (16384, 2)
        np.save(os.path.join(HERE, "coe_cpp_14_12.npy"), sc.astype(np.int16))
        E["coe_cpp_14_12"] = {"source": "reference", "what": "coe.dat written by main() of cpp/cordic_sincos.cpp (lines: 's c')",
                              "text_md5": hashlib.md5(raw).hexdigest(), "file": "coe_cpp_14_12.npy",
                              "params": pdict(O.oparams(1, 14, 12, model=O.MODEL_CPP))}
        assert E["coe_cpp_14_12"]["text_md5"] == "b65f091fb2afeeb252aa0bc5728fe46a", "differs from SURVEY App. B"
        s, c = O.sincos(O.oparams(1, 14, 12, model=O.MODEL_CPP), 0, 16384)
        assert np.array_equal(s, sc[:, 0]) and np.array_equal(c, sc[:, 1]), "oracle model A != coe.dat"

    # ---- reference: cordic() of cpp/cordic_sincos.cpp at other widths (oracle/_ref) -------------------
    rng = np.random.default_rng(20240917)
    for (pw, w), path in sorted(ref.items()):
        n = 1 << pw
        if pw <= 12:
            th = np.arange(n, dtype=np.int64)
        else:
            edges = np.concatenate([q * (n // 4) + np.arange(-40, 40) for q in range(5)])
            th = np.unique(np.concatenate([np.arange(512), edges, rng.integers(0, n, 1500)]) % n)
        s, c = O.RefCordic(path).sweep(th)
        os_, oc = [], []
        p = O.oparams(1, pw, w, model=O.MODEL_CPP)
        for t in th:
            a, b = O.sincos(p, int(t), 1)
            os_.append(a[0]); oc.append(b[0])
        assert np.array_equal(s, np.array(os_)) and np.array_equal(c, np.array(oc)), f"oracle model A != _ref at {pw}/{w}"
        name = f"sincos_cpp_{pw}_{w}"
        np.save(os.path.join(HERE, name + ".npy"), np.stack([th.astype(np.int64), s.astype(np.int64), c.astype(np.int64)]))
        E[name] = {"source": "reference", "what": "rows: theta, sin, cos from cordic() of cpp/cordic_sincos.cpp",
                   "file": name + ".npy", "params": pdict(p)}

    # ---- survey App. B known answers for the HLS model -------------------------------------------------
    def survey(name, p, n0, count, expect, keep=False, sparse=None):
        a = O.generate(p, n0, count)
        st = stats(a)
        for k, v in expect.items():
            assert st[k] == v, f"{name}: oracle {k}={st[k]} != SURVEY {v}"
        if sparse:
            for n, v in sparse.items():
                assert int(O.generate(p, n, 1)[0]) == v, f"{name}: sample {n}"
        e = {"source": "survey", "params": pdict(p), "n0": n0, **st}
        if sparse:
            e["sparse"] = {str(k): v for k, v in sparse.items()}
        if keep:
            np.save(os.path.join(HERE, name + ".npy"), a)
            e["file"] = name + ".npy"
        E[name] = e

    survey("C1_hamming_12_16", O.oparams(1, 12, 16), 0, 4096,
           {"md5": "3c8b72bf87d0aab8c5530a19992427c7", "sum": 72943615, "min": 2850, "max": 32767}, keep=True,
           sparse={0: 2850, 1000: 17260, 2048: 32767, 4095: 2850})
    survey("C2_bh4_20_24", O.oparams(4, 20, 24), 0, 1 << 20,
           {"md5": "647ea2f75377de962da29d5ab6848764", "sum": 3155598770172, "min": 503, "max": 8388607},
           sparse={0: 503, 1: 504, 262144: 1824272, 524288: 8388607, 1048575: 504})
    survey("C4_bh4_16_24_frame", O.oparams(4, 16, 24), 0, 1 << 16,
           {"md5": "5ad5e625609e89573d2bd8529d5eef6a", "sum": 197224923136, "min": 503, "max": 8388607})
    survey("bh5_10_24", O.oparams(5, 10, 24), 0, 1 << 10,
           {"md5": "46e784b28a4e1a74fb90e8cc13978b4c", "sum": 1388198916, "min": 91, "max": 4194303}, keep=True)
    survey("bh7_4_16", O.oparams(7, 4, 16), 0, 16, {"sum": 71099, "min": 0, "max": 16381}, keep=True)
    assert list(O.generate(O.oparams(7, 4, 16), 0, 16)) == [0, 1, 19, 184, 1046, 3653, 8518, 13938, 16381, 13938, 8518,
                                                              3653, 1046, 184, 19, 1]
    # sin/cos of hls/cordic at 10/16 (SURVEY App. B sparse samples: (cos, sin))
    p = O.oparams(1, 10, 16)
    sp = {0: (16383, 0), 1: (16383, 100), 2: (16382, 201), 255: (100, 16383), 256: (0, 16383), 257: (-100, 16383),
          512: (-16383, 0), 768: (0, -16383), 1023: (16383, -100)}
    for th, (c, s) in sp.items():
        ss, cc = O.sincos(p, th, 1)
        assert (int(cc[0]), int(ss[0])) == (c, s), f"sincos hls 10/16 theta {th}"
    E["sincos_hls_10_16_sparse"] = {"source": "survey", "params": pdict(p), "cos_sin": {str(k): list(v) for k, v in sp.items()}}

    # C3/C5: BH-7, 26/32, eight shards of 2^23 (SURVEY App. B shard sums, global min/max, sparse samples)
    sums = [1459592223587, 163245881214546, 2222536257447441, 7384501030013110, 7384501544450686,
            2222536745594365, 163245948154191, 1459593090495]
    with Pool(8) as pool:
        shards = dict(pool.map(c3_shard, range(8)))
    for g in range(8):
        assert shards[g]["sum"] == sums[g], f"C3 shard {g} sum"
    assert min(s["min"] for s in shards.values()) == 65 and max(s["max"] for s in shards.values()) == 1073741825
    p = O.oparams(7, 26, 32)
    sparse = {0: 68, 1: 68, 1 << 20: 474, (1 << 23) - 1: 1104462, 1 << 24: 68681275, (1 << 25) - 1: 1073741822,
              1 << 25: 1073741825, (1 << 25) + 1: 1073741822, 3 << 24: 68681275, (1 << 26) - 1: 68}
    for n, v in sparse.items():
        assert int(O.generate(p, n, 1)[0]) == v
    E["C3_bh7_26_32"] = {"source": "survey", "what": "shard g = [g*2^23, (g+1)*2^23); sums/min/max/sparse from SURVEY App. B, "
                         "md5/fnv/strided samples from the oracle that reproduces them", "params": pdict(p),
                         "shards": [shards[g] for g in range(8)], "sparse": {str(k): v for k, v in sparse.items()}}

    # the same headline window with the cpp model's cosines (model A is the one pinned by oracle/_ref)
    with Pool(8) as pool:
        shards_cpp = dict(pool.map(c3_shard_cpp, range(8)))
    pc = O.oparams(7, 26, 32, model=O.MODEL_CPP)
    E["C3cpp_bh7_26_32"] = {"source": "reference", "note": "BH-7 2^26/32-bit: six calls of the reference's compiled cordic() "
                            "(oracle/_ref, cpp/cordic_sincos.cpp built at PHASE_WIDTH 26 / DATA_WIDTH 32) per coefficient in the "
                            "cosine-sum of hls/windows/win_function.cpp:361-375; per-shard checksums",
                            "params": pdict(pc), "shards": [shards_cpp[g] for g in range(8)]}
    # the oracle's restatement of model A gives the same shards (cross-check, one shard)
    chk = O.generate_mt(pc, 5 << 23, 1 << 23)
    assert md5(chk) == shards_cpp[5]["md5"], "oracle model A != reference-compiled window"

    # ---- oracle-only vectors: no runnable reference (parity unpinned) or derived configurations --------
    def oracle_only(name, p, n0, count, keep=True, note="parity unpinned: restated from the VHDL, no simulator here"):
        a = O.generate(p, n0, count)
        e = {"source": "oracle", "note": note, "params": pdict(p), "n0": n0, **stats(a)}
        if keep:
            np.save(os.path.join(HERE, name + ".npy"), a)
            e["file"] = name + ".npy"
        E[name] = e

    tb7 = [round(v * (2 ** 23 - 1)) for v in (0.271220360585039, 0.433444612327442, 0.218004122892930, 0.065785343295606,
                                               0.010761867305342, 0.000770012710581, 0.000013680883060)]
    oracle_only("vhdl_hamming_11_16", O.oparams(1, 11, 16, model=O.MODEL_VHDL, combine=O.COMBINE_VHDL), 0, 2048)
    oracle_only("vhdl_bh7_10_24", O.oparams(7, 10, 24, model=O.MODEL_VHDL, combine=O.COMBINE_VHDL, aa=tb7), 0, 1024)
    oracle_only("vhdl_bh4_12_24_p3", O.oparams(4, 12, 24, model=O.MODEL_VHDL, combine=O.COMBINE_VHDL, precision=3), 0, 4096)
    oracle_only("taylor_hamming_12_16_l9", O.oparams(1, 12, 16, combine=O.COMBINE_VHDL, sin_type=O.SIN_TAYLOR, lut_size=9), 0, 4096)
    oracle_only("taylor_bh3_14_24_l9", O.oparams(3, 14, 24, combine=O.COMBINE_VHDL, sin_type=O.SIN_TAYLOR, lut_size=9), 0, 16384, keep=False)
    oracle_only("cpp_bh7_12_32_hlscombine", O.oparams(7, 12, 32, model=O.MODEL_CPP), 0, 4096,
                note="model A cosines (pinned by oracle/_ref) in the HLS cosine-sum rule")

    # ---- the all-term-count Taylor extension and the variant generators (all oracle-only) ----
    ext = "extension BHW_SIN_TAYLOR_ALL (include/bhw.h): no reference counterpart, the oracle defines it"
    oracle_only("taylor_all_bh7_12_24_l9", O.oparams(7, 12, 24, combine=O.COMBINE_VHDL, sin_type=O.SIN_TAYLOR_ALL, lut_size=9),
                0, 4096, keep=False, note=ext)
    oracle_only("taylor_all_bh5_13_16_l9", O.oparams(5, 13, 16, sin_type=O.SIN_TAYLOR_ALL, lut_size=9), 0, 8192, keep=False, note=ext)
    oracle_only("taylor_all_bh4_14_32_l10", O.oparams(4, 14, 32, combine=O.COMBINE_VHDL, sin_type=O.SIN_TAYLOR_ALL, lut_size=10),
                100, 16384, keep=False, note=ext)
    for name, model, pw, w in (("sincos_dds48_12_16", O.MODEL_DDS48, 12, 16), ("sincos_dds48_14_32", O.MODEL_DDS48, 14, 32),
                               ("sincos_scaled_12_16", O.MODEL_SCALED, 12, 16), ("sincos_scaled_14_24", O.MODEL_SCALED, 14, 24),
                               ("sincos_scaled_20_10", O.MODEL_SCALED, 20, 10)):
        po = O.oparams(1, pw, w, model=model)
        cnt = min(1 << pw, 16384)
        sn, cs = O.sincos(po, (1 << pw) - cnt // 2 if pw > 14 else 0, cnt)
        E[name] = {"source": "oracle", "note": "parity unpinned: restated from src/cordic_dds48.vhd / src/cordic_dds_scaled.vhd",
                   "params": pdict(po), "theta0": (1 << pw) - cnt // 2 if pw > 14 else 0, "count": cnt,
                   "sin_md5": md5(sn), "cos_md5": md5(cs)}
    rng2 = np.random.default_rng(20241003)
    for P, IW, AW in ((1, 23, 24), (3, 16, 16), (4, 32, 32)):
        lo, hi = -(1 << (IW - 1)), (1 << (IW - 1)) - 1
        x = rng2.integers(lo, hi + 1, 1500)
        y = rng2.integers(lo, hi + 1, 1500)
        phi = O.atan2(P, IW, AW, x, y)
        name = f"atan2_p{P}_{IW}_{AW}"
        np.save(os.path.join(HERE, name + ".npy"), np.stack([x.astype(np.int64), y.astype(np.int64), phi.astype(np.int64)]))
        E[name] = {"source": "oracle", "note": "parity unpinned: restated from src/cordic_atan2.vhd; rows: VEC_DX, VEC_DY, PHI_DT",
                   "file": name + ".npy", "precision": P, "input_width": IW, "angle_width": AW}

    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", len(E), "entries")


if __name__ == "__main__":
    main()
