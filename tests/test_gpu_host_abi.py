"""Host-side behaviour of the C ABI on a GPU (include/bhw.h "Threading", bhw_prepare_device, packed-format verification)."""
import ctypes
import threading

import numpy as np
import pytest

import oracle_lib as O
from blackman_harris_win_amd import binding as B

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def test_two_threads_same_stream_share_library_scratch_safely(torch):
    """Two host threads, the same (device, stream), different windows, library-owned scratch: the launches of one call must
    not interleave with the other's (A.build, B.build, A.combine would combine A from B's table), and a call that needs a larger
    buffer must not free the one the other is about to launch with."""
    import blackman_harris_win_amd as bhw
    pa = B.make_params(7, 18, 32)
    pb = B.make_params(5, 20, 24, model=B.MODEL_CPP)                  # larger table: forces a re-allocation of the shared slot
    pc = B.make_params(4, 16, 30, model=B.MODEL_VHDL, combine=B.COMBINE_VHDL)
    want = {id(p): O.generate_mt(O.from_bhw(p), 0, 1 << p.phi_width) for p in (pa, pb, pc)}
    st = torch.cuda.Stream()
    errors = []

    def worker(p, rounds):
        try:
            with torch.cuda.stream(st):
                for _ in range(rounds):
                    out = bhw.generate(p, 0, 1 << p.phi_width, algo=B.ALGO_TABLE)
                    st.synchronize()
                    if not np.array_equal(out.cpu().numpy(), want[id(p)]):
                        errors.append(("mismatch", p.n_terms, p.phi_width))
                        return
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(p, 40)) for p in (pa, pb, pc)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    # the NULL-stream helper shares one slot between threads as well
    host = [(ctypes.c_int32 * (1 << p.phi_width))() for p in (pa, pc)]

    def to_host(p, buf):
        for _ in range(10):
            rc = B.lib().bhw_generate_to_host(ctypes.byref(p), 0, 0, 1 << p.phi_width, buf)
            if rc != 0 or not np.array_equal(np.ctypeslib.as_array(buf), want[id(p)]):
                errors.append(("to_host", rc))
                return

    threads = [threading.Thread(target=to_host, args=(p, b)) for p, b in zip((pa, pc), host)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_current_device_is_restored_and_errors_are_not_consumed(torch):
    """The entry points leave the thread's current device alone and neither read nor clear its hipGetLastError() state."""
    hip = ctypes.CDLL("libamdhip64.so")
    cur = ctypes.c_int(-1)
    assert hip.hipGetDevice(ctypes.byref(cur)) == 0
    before = cur.value
    p = B.make_params(4, 12, 24)
    out = torch.empty(4096, dtype=torch.int32, device="cuda")
    # leave an error behind on this thread (an invalid device ordinal), as another library might
    assert hip.hipSetDevice(12345) != 0
    assert B.lib().bhw_generate_device(ctypes.byref(p), 0, None, 0, 4096, ctypes.c_void_p(out.data_ptr())) == 0
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), O.generate(O.from_bhw(p), 0, 4096))
    assert hip.hipGetDevice(ctypes.byref(cur)) == 0 and cur.value == before
    assert hip.hipGetLastError() != 0          # still there: the call did not swallow it (and did not fail because of it)
    assert hip.hipGetLastError() == 0


def test_prepare_then_graph_capture_without_workspace(torch):
    """After bhw_prepare_device a table call with library scratch and a Taylor call neither allocate nor synchronise: both can
    be captured into a HIP graph; unprepared, the Taylor call refuses inside a capture instead of breaking it."""
    import blackman_harris_win_amd as bhw
    st = torch.cuda.Stream()
    pt = B.make_params(7, 22, 30)
    # lut_size 14: a ROM no other test uploads (the fuzz slice draws at most 13)
    py = B.make_params(3, 20, 29, combine=B.COMBINE_VHDL, sin_type=B.SIN_TAYLOR, lut_size=14)
    want_t = O.generate_mt(O.from_bhw(pt), 0, 1 << 22)
    with torch.cuda.stream(st):
        out_t = torch.zeros(1 << 22, dtype=torch.int32, device="cuda")
        out_y = torch.zeros(1 << 15, dtype=torch.int32, device="cuda")
        st.synchronize()
        g_bad = torch.cuda.CUDAGraph()
        with pytest.raises(B.BhwError) as ei:
            with torch.cuda.graph(g_bad, stream=st):
                bhw.generate(py, 777, 1 << 15, out=out_y)
        assert ei.value.code == -3 and "bhw_prepare_device" in ei.value.detail
        del g_bad
        bhw.prepare(pt)
        bhw.prepare(py)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            bhw.generate(pt, 0, 1 << 22, out=out_t)
            bhw.generate(py, 777, 1 << 15, out=out_y)
        for _ in range(2):
            out_t.zero_()
            out_y.zero_()
            graph.replay()
            st.synchronize()
            assert np.array_equal(out_t.cpu().numpy(), want_t)
            assert np.array_equal(out_y.cpu().numpy(), O.generate_mt(O.from_bhw(py), 777, 1 << 15))


def test_prepare_settles_the_escape_format_for_capture(torch):
    """The cpp model at 2^24 / 32 bits: plain nibbles are refused, nibble + escapes holds.  bhw_prepare_device settles both verdicts,
    so a captured call builds the escape tables inside the graph (no read-back) and its replays match the oracle."""
    import blackman_harris_win_amd as bhw
    p = B.make_params(7, 24, 32, model=B.MODEL_CPP)
    n = 1 << 24
    want = O.generate_mt(O.from_bhw(p), 0, n)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        out = torch.zeros(n, dtype=torch.int32, device="cuda")
        bhw.prepare(p)
        assert B.describe_plan(p, 0, n, algo=B.ALGO_TABLE).startswith("table[nibble+esc]:"), B.describe_plan(p, 0, n, algo=B.ALGO_TABLE)
        st.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            bhw.generate(p, 0, n, out=out, algo=B.ALGO_TABLE)
        for _ in range(2):
            out.zero_()
            graph.replay()
            st.synchronize()
            assert np.array_equal(out.cpu().numpy(), want)


def test_prepare_covers_table_calls_of_fused_size_windows(torch):
    """Round-3 advisor finding: bhw_prepare_device skipped the scratch when AUTO sends the WHOLE period to the fused kernel, yet a
    partial range of such a window (no whole period -> table strategy) or an explicit BHW_ALGO_TABLE still builds a table; inside
    a stream capture those calls then failed in the allocator.  After prepare both capture and replay bit-exactly."""
    import blackman_harris_win_amd as bhw
    st = torch.cuda.Stream()
    p = B.make_params(4, 20, 23)                                   # BH-4 2^20: AUTO = fused for whole periods
    n = 1 << 20
    assert B.describe_plan(p, 0, n).startswith("fused") and B.describe_plan(p, 0, n // 2).startswith("table")
    want = O.generate_mt(O.from_bhw(p), 0, n)
    with torch.cuda.stream(st):
        half = torch.zeros(n // 2, dtype=torch.int32, device="cuda")
        whole = torch.zeros(n, dtype=torch.int32, device="cuda")
        bhw.prepare(p)
        st.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            bhw.generate(p, n // 4, n // 2, out=half)                        # partial range: table strategy, library scratch
            bhw.generate(p, 0, n, out=whole, algo=B.ALGO_TABLE)              # explicit table strategy, no workspace
        for _ in range(2):
            half.zero_()
            whole.zero_()
            graph.replay()
            st.synchronize()
            assert np.array_equal(half.cpu().numpy(), want[n // 4:3 * n // 4])
            assert np.array_equal(whole.cpu().numpy(), want)


def test_library_scratch_is_sized_by_the_format_in_use(torch):
    """The library-owned scratch of a stream holds the table format the configuration actually uses (16.5 MiB for the 2^26 / 32-bit
    window with nibble entries), not the 8-bytes-per-entry bound; a caller's workspace of bhw_workspace_bytes_ex bytes is accepted."""
    import blackman_harris_win_amd as bhw
    p = B.make_params(7, 24, 32)
    n = 1 << 24
    bhw.prepare(p)                                                   # settles the packed formats of (HLS, 24, 32)
    ex = B.BhwExec()
    ex.struct_size = ctypes.sizeof(B.BhwExec)
    ex.algo = B.ALGO_TABLE
    tight = B.lib().bhw_workspace_bytes_ex(ctypes.byref(p), 0, n, ctypes.byref(ex))
    bound = B.lib().bhw_workspace_bytes(ctypes.byref(p), 0, n, B.ALGO_TABLE)
    assert "nibble" in B.describe_plan(p, 0, n, algo=B.ALGO_TABLE) and tight < bound // 7
    ws = torch.empty(tight, dtype=torch.uint8, device="cuda")
    out = bhw.generate(p, 0, n, algo=B.ALGO_TABLE, workspace=ws)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), O.generate_mt(O.from_bhw(p), 0, n))
    with pytest.raises(B.BhwError) as ei:
        bhw.generate(p, 0, n, algo=B.ALGO_TABLE, workspace=ws[:tight - 256])
    assert ei.value.code == -4


def test_library_scratch_gives_the_excess_back_once_the_formats_are_settled(torch):
    """Round-4 advisor finding: the first call of a configuration has to size the stream's scratch for every format it may try (8 bytes
    per entry: 128 MiB for the 2^26-point window at 32 bits) and the slot kept that size for good.  Now the next call of the
    configuration re-sizes it to the format in use (16.5 MiB), bhw_prepare_device leaves it there directly, and every later call
    produces the same window.  After prepare a captured call with an explicit, wider table_format is refused with a message that
    names the remedy (the library scratch cannot grow inside a capture) and works with the caller's workspace."""
    import blackman_harris_win_amd as bhw
    L = _dbg()
    L.bhw_dbg_library_scratch_bytes.restype = ctypes.c_uint64
    L.bhw_dbg_library_scratch_bytes.argtypes = [ctypes.c_int, ctypes.c_void_p]
    p = B.make_params(7, 26, 32)
    n = 1 << 26
    d = ctypes.c_uint32()
    assert L.bhw_dbg_table_format_info(ctypes.byref(p), ctypes.byref(d), None) == 0 and d.value == 9
    ex = B.BhwExec()
    ex.struct_size = ctypes.sizeof(B.BhwExec)
    ex.algo = B.ALGO_TABLE
    bound = B.lib().bhw_workspace_bytes(ctypes.byref(p), 0, n, B.ALGO_TABLE)
    for k in (16 + 9, 48 + 9, 9, 6):
        assert L.bhw_dbg_table_format_verdict(ctypes.byref(p), k, 3) == 0              # forget what earlier tests settled
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        out = torch.zeros(n, dtype=torch.int32, device="cuda")
        bhw.generate(p, 0, n, out=out)                                                   # unprepared: tries the formats
        st.synchronize()
        assert L.bhw_dbg_library_scratch_bytes(0, ctypes.c_void_p(st.cuda_stream)) == bound
        first = out.clone()
        tight = B.lib().bhw_workspace_bytes_ex(ctypes.byref(p), 0, n, ctypes.byref(ex))
        assert tight < bound // 7
        out.zero_()
        bhw.generate(p, 0, n, out=out)                                                   # verdicts known: the excess goes back
        st.synchronize()
        assert L.bhw_dbg_library_scratch_bytes(0, ctypes.c_void_p(st.cuda_stream)) == tight
        assert torch.equal(out, first)
    st2 = torch.cuda.Stream()
    with torch.cuda.stream(st2):
        bhw.prepare(p)
        assert L.bhw_dbg_library_scratch_bytes(0, ctypes.c_void_p(st2.cuda_stream)) == tight
        for k in (16 + 9, 9, 6):
            assert L.bhw_dbg_table_format_verdict(ctypes.byref(p), k, 0) in (1, 2)      # every explicit choice has its verdict
        out2 = torch.zeros(n, dtype=torch.int32, device="cuda")
        st2.synchronize()
        g_bad = torch.cuda.CUDAGraph()
        with pytest.raises(B.BhwError) as ei:
            with torch.cuda.graph(g_bad, stream=st2):
                bhw.generate(p, 0, n, out=out2, algo=B.ALGO_TABLE, table_format=B.TABLE_RESIDUAL)
        assert ei.value.code == -3 and "workspace" in ei.value.detail
        del g_bad
        ex.table_format = B.TABLE_RESIDUAL
        ws = torch.empty(B.lib().bhw_workspace_bytes_ex(ctypes.byref(p), 0, n, ctypes.byref(ex)), dtype=torch.uint8, device="cuda")
        assert tight < ws.numel() < bound
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st2):
            bhw.generate(p, 0, n, out=out2, algo=B.ALGO_TABLE, table_format=B.TABLE_RESIDUAL, workspace=ws)
        graph.replay()
        st2.synchronize()
        assert torch.equal(out2, first)
    # an oversized slot met by a captured call: nothing is re-sized inside the capture (that would synchronise the stream), the graph
    # uses the buffer as it is, and the next plain call gives the excess back
    for k in (16 + 9, 48 + 9, 9, 6):
        assert L.bhw_dbg_table_format_verdict(ctypes.byref(p), k, 3) == 0
    st3 = torch.cuda.Stream()
    with torch.cuda.stream(st3):
        out3 = torch.zeros(n, dtype=torch.int32, device="cuda")
        bhw.generate(p, 0, n, out=out3)                                                  # unprepared again: the slot of st3 takes the bound
        st3.synchronize()
        assert L.bhw_dbg_library_scratch_bytes(0, ctypes.c_void_p(st3.cuda_stream)) == bound
        out3.zero_()
        g3 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g3, stream=st3):
            bhw.generate(p, 0, n, out=out3)
        assert L.bhw_dbg_library_scratch_bytes(0, ctypes.c_void_p(st3.cuda_stream)) == bound
        g3.replay()
        st3.synchronize()
        assert torch.equal(out3, first)
        del g3                                                                           # (the graph's kernels hold the buffer's address)
        out3.zero_()
        bhw.generate(p, 0, n, out=out3)
        st3.synchronize()
        assert L.bhw_dbg_library_scratch_bytes(0, ctypes.c_void_p(st3.cuda_stream)) == tight
        assert torch.equal(out3, first)


def _dbg():
    L = B.lib()
    P = ctypes.POINTER(B.BhwParams)
    L.bhw_dbg_check_table_format.argtypes = [P, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p,
                                             ctypes.POINTER(ctypes.c_uint32)]
    L.bhw_dbg_table_format_verdict.argtypes = [P, ctypes.c_uint32, ctypes.c_int]
    L.bhw_dbg_table_format_info.argtypes = [P, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    return L


def test_packed_format_overflow_is_detected_on_the_device(torch):
    """The build kernels' check word: clear for the formats the library would choose, set when a format is forced onto a
    configuration it cannot hold (delta16 at W - PW = 10: the drift over a 64-entry block exceeds int16; residual cells of
    2^9 entries at 2^22 / 32 bits: the curvature alone exceeds int8)."""
    L = _dbg()
    flag = ctypes.c_uint32(7)
    for win, pw, w, model, dlog, expect in [(7, 26, 32, B.MODEL_HLS, 9, 0), (7, 26, 32, B.MODEL_CPP, 9, 0), (7, 24, 32, B.MODEL_HLS, 6, 0),
                                            (7, 22, 32, B.MODEL_HLS, 6, 1), (7, 22, 32, B.MODEL_HLS, 9, 1), (4, 22, 24, B.MODEL_VHDL, 9, 0),
                                            # 4-bit fields (16 + d): the HLS model's deviations stay within -5 .. 6, the cpp model's reach 10
                                            (7, 26, 32, B.MODEL_HLS, 16 + 9, 0), (7, 26, 32, B.MODEL_CPP, 16 + 9, 1),
                                            # ... which the escape tables (48 + d) hold: 547 entries, at most 36 per build workgroup; cells of
                                            # 2^9 entries at 2^24 / 32 bits put the curvature on top and overflow the tables
                                            (7, 26, 32, B.MODEL_CPP, 48 + 9, 0), (7, 26, 32, B.MODEL_VHDL, 48 + 9, 0), (7, 24, 32, B.MODEL_CPP, 48 + 9, 1)]:
        p = B.make_params(win, pw, w, model=model)
        ws = torch.empty(B.lib().bhw_workspace_bytes(ctypes.byref(p), 0, 1 << pw, B.ALGO_TABLE), dtype=torch.uint8, device="cuda")
        assert L.bhw_dbg_check_table_format(ctypes.byref(p), 0, None, dlog, ctypes.c_void_p(ws.data_ptr()), ctypes.byref(flag)) == 0
        assert flag.value == expect, (win, pw, w, model, dlog)


def test_overflowing_format_falls_back_and_stays_exact(torch):
    """A configuration whose packed-format verdict is 'overflows' is generated through the next wider format, bit-exactly."""
    import blackman_harris_win_amd as bhw
    L = _dbg()
    p = B.make_params(5, 22, 26, model=B.MODEL_VHDL, precision=2)         # a configuration no other test touches
    d, ok16 = ctypes.c_uint32(), ctypes.c_uint32()
    assert L.bhw_dbg_table_format_info(ctypes.byref(p), ctypes.byref(d), ctypes.byref(ok16)) == 0
    assert d.value >= 7 and ok16.value == 1
    nib = 16 + d.value                                                                   # the 4-bit form is tried first
    for k in (nib, 48 + d.value, d.value, 6):                                            # unknown so far (the fuzz tests draw random
        assert L.bhw_dbg_table_format_verdict(ctypes.byref(p), k, 3) == 0                # configurations: forget what they may have settled)
    want = O.generate_mt(O.from_bhw(p), 0, 1 << 22)
    assert np.array_equal(bhw.generate(p, 0, 1 << 22, algo=B.ALGO_TABLE).cpu().numpy(), want)
    v_nib = L.bhw_dbg_table_format_verdict(ctypes.byref(p), nib, 0)
    assert v_nib in (1, 2)                                                               # decided on first use, either way
    if v_nib == 1:
        assert L.bhw_dbg_table_format_verdict(ctypes.byref(p), nib, 2) == 2              # pretend it overflowed
        assert np.array_equal(bhw.generate(p, 0, 1 << 22, algo=B.ALGO_TABLE).cpu().numpy(), want)   # 4-bit fields + escape tables now
    esc = 48 + d.value
    v_esc = L.bhw_dbg_table_format_verdict(ctypes.byref(p), esc, 0)
    assert v_esc in (1, 2)                                                               # decided when the plain nibbles were refused
    if v_esc != 2:
        assert L.bhw_dbg_table_format_verdict(ctypes.byref(p), esc, 2) == 2              # pretend its tables overflowed
    assert np.array_equal(bhw.generate(p, 0, 1 << 22, algo=B.ALGO_TABLE).cpu().numpy(), want)       # 8-bit fields now
    assert L.bhw_dbg_table_format_verdict(ctypes.byref(p), d.value, 0) == 1              # verified exact on first use
    assert L.bhw_dbg_table_format_verdict(ctypes.byref(p), d.value, 2) == 2              # pretend it overflowed
    assert np.array_equal(bhw.generate(p, 0, 1 << 22, algo=B.ALGO_TABLE).cpu().numpy(), want)   # delta16 now
    assert L.bhw_dbg_table_format_verdict(ctypes.byref(p), 6, 0) == 1
    assert L.bhw_dbg_table_format_verdict(ctypes.byref(p), 6, 2) == 2
    assert np.array_equal(bhw.generate(p, 0, 1 << 22, algo=B.ALGO_TABLE).cpu().numpy(), want)   # plain now
    assert L.bhw_dbg_table_format_verdict(ctypes.byref(p), d.value, 1) == 1
    assert L.bhw_dbg_table_format_verdict(ctypes.byref(p), 6, 1) == 1
    assert L.bhw_dbg_table_format_verdict(ctypes.byref(p), nib, v_nib) == v_nib
    assert L.bhw_dbg_table_format_verdict(ctypes.byref(p), esc, v_esc) == v_esc


def test_best_format_per_model(torch):
    """BEST resolves to one byte per entry for the HLS model of the headline window, and to one byte plus the escape tables for
    the cpp model, whose deviations do not all fit four bits (decided by the device-side check of the first build, then cached)."""
    import blackman_harris_win_amd as bhw
    for model, want in ((B.MODEL_HLS, "table[nibble]"), (B.MODEL_CPP, "table[nibble+esc]")):
        p = B.make_params(7, 26, 32, model=model)
        bhw.generate(p, 0, 1 << 26, algo=B.ALGO_TABLE)
        assert B.describe_plan(p, 0, 1 << 26, algo=B.ALGO_TABLE).startswith(want), B.describe_plan(p, 0, 1 << 26, algo=B.ALGO_TABLE)


def test_out_tensors_are_validated(torch):
    import blackman_harris_win_amd as bhw
    p = B.make_params(4, 10, 24)
    x = torch.zeros(1024, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        bhw.apply(p, x, out=torch.zeros(1000, dtype=torch.int32, device="cuda"))          # too short
    with pytest.raises(ValueError):
        bhw.apply(p, x, out=torch.zeros(2048, dtype=torch.int32, device="cuda")[::2])     # strided
    with pytest.raises(ValueError):
        bhw.generate_batched(p, 4, out=torch.zeros(4 * 1024, dtype=torch.int64, device="cuda"))
    with pytest.raises(ValueError):
        bhw.generate_batched(p, 4, out=torch.zeros(3 * 1024, dtype=torch.int32, device="cuda"))
    with pytest.raises(ValueError):
        bhw.generate(p, 0, 1024, out=torch.zeros(1024, dtype=torch.int32))                # not on the GPU


@pytest.mark.parametrize("pw", [28, 30])
def test_long_windows_27_to_30_bits(torch, pw):
    """phi_width above the reference's documented 26 (README.md:2): the same arithmetic on a longer counter.  Whole 2^28 /
    2^30 windows through the tile path, against DIRECT slices (device-side comparison) and the oracle on sampled indices."""
    import blackman_harris_win_amd as bhw
    w = 32
    p = B.make_params(7, pw, w)
    n = 1 << pw
    full = bhw.generate(p, 0, n, algo=B.ALGO_TABLE)
    rng = np.random.default_rng(pw)
    starts = [0, n // 8 - 3000, n // 4 - 3000, n // 2 - 3000, 3 * (n // 4) - 3000, n - 6000] + [int(v) for v in rng.integers(0, n - 6000, 6)]
    for s0 in starts:
        sl = bhw.generate(p, s0, 6000, algo=B.ALGO_DIRECT)
        assert bool((sl == full[s0:s0 + 6000]).all()), s0
    idx = np.unique(np.concatenate([rng.integers(0, n, 600), [0, 1, n // 2, n // 2 - 1, n - 1]]))
    po = O.from_bhw(p)
    want = np.array([O.generate(po, int(i), 1)[0] for i in idx], dtype=np.int32)
    got = full[torch.as_tensor(idx, device="cuda")].cpu().numpy()
    assert np.array_equal(got, want)
    # quadrant fold as a size-independent property: the window sum and its extremes sit where they must
    assert int(full[n // 2]) == int(full.max())
    del full
    B.lib().bhw_release_device(0)
