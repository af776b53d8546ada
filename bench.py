#!/usr/bin/env python3
"""Headline benchmark: window Gsamples/s + fraction of the HBM-write roofline,
Blackman-Harris 7-term, N = 2^26 (64M points), 32-bit output (BASELINE.json configs[2]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Without WORLD_SIZE in the environment `python bench.py --gpus N` (N > 1) starts its N ranks itself (a torch.distributed.run child
process, before this process touches a GPU).  One process per GPU.  A step = one pass of the hot path over one batch: 2^26 coefficients generated
from the parameter set into a resident HBM buffer (no inputs; nothing cached between steps: the
shared CORDIC table is rebuilt inside every step).  With N ranks the coefficient stream is sharded by
contiguous index range -- rank r produces stream indices [r*2^26, (r+1)*2^26) -- with no data-path
collective (weak scaling: fixed work per GPU; the default, "scaling": "weak").
`--scaling strong` is BASELINE config C5 instead: ONE 2^26 window over the N ranks, each producing its interleaved
ownership part (bhw_generate_part_device) into a full-length buffer, still without a collective; value is then
2^26 coefficients x steps / time.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PHI_WIDTH, DAT_WIDTH, WIN = 26, 32, 7
COUNT = 1 << PHI_WIDTH
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3-6.9 achievable)
BYTES_PER_COEFF = 4          # 4 bytes written per coefficient, 0 read


def shard_for(rank):
    """Contiguous stream-index shard of this rank (weak scaling: COUNT per rank)."""
    return rank * COUNT, COUNT


def timed_steps(step_fn, steps, warmup, barrier, sync, allreduce_max):
    """The measurement protocol of the bench contract: W untimed steps, then exactly K steps bracketed by
    barrier + device sync on both sides; returns the MAX over ranks of the elapsed seconds."""
    for _ in range(warmup):
        step_fn()
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync()
    barrier()
    t1 = time.perf_counter()
    return allreduce_max(t1 - t0)


def cpu_baseline(target_seconds=12.0, max_threads=16):
    """The reference's own cordic() (oracle/_ref, built from cpp/cordic_sincos.cpp) timed on this host's
    cores on a bounded sample of the same workload; falls back to the oracle port when _ref is absent."""
    import numpy as np
    import oracle_lib as O
    so = os.path.join(ROOT, "oracle", "libcpubaseline.so")
    if not os.path.exists(so):
        return None
    L = ctypes.CDLL(so)
    L.bhw_cpu_baseline.restype = ctypes.c_double
    L.bhw_cpu_baseline.argtypes = [ctypes.c_char_p, ctypes.POINTER(O.OParams), ctypes.c_uint64, ctypes.c_uint64,
                                   ctypes.c_int, ctypes.c_void_p]
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(avail, max_threads))   # the GPU box's CPU share for one GPU is 16 cores
    ref = os.path.join(ROOT, "oracle", "_ref", f"libref_cordic_{PHI_WIDTH}_{DAT_WIDTH}.so")
    kind = "reference" if os.path.exists(ref) else "port"
    # the reference cordic() is model CPP (cpp/cordic_sincos.cpp); same loop and widths as the HLS model
    p = O.oparams(WIN, PHI_WIDTH, DAT_WIDTH, model=O.MODEL_CPP if kind == "reference" else O.MODEL_HLS)
    lib = ref.encode() if kind == "reference" else None
    n0 = COUNT // 4 - (1 << 16)
    probe = 1 << 17
    buf = np.empty(probe, np.int32)
    dt = L.bhw_cpu_baseline(lib, ctypes.byref(p), n0, probe, threads, buf.ctypes.data)
    if dt <= 0:
        return None
    count = int(min(COUNT, max(probe, probe * target_seconds / dt)))
    buf = np.empty(count, np.int32)
    dt = L.bhw_cpu_baseline(lib, ctypes.byref(p), n0, count, threads, buf.ctypes.data)
    # sanity: the harness around the reference cordic() agrees with the oracle on the head of the sample
    ok = bool(np.array_equal(buf[:2048], O.generate(p, n0, 2048)))
    what = ("6 calls of cordic() of cpp/cordic_sincos.cpp (compiled from the reference source at PHASE_WIDTH 26 / "
            "DATA_WIDTH 32) + HLS cosine-sum per coefficient") if kind == "reference" else "oracle C restatement (model HLS)"
    return {"value": count / dt / 1e9, "unit": "Gsamples/s", "cores": threads, "kind": kind,
            "sample": f"{count} coefficients of the same BH-7 2^26/32-bit window from n0={n0}, {dt:.2f} s wall, {what}",
            "matches_oracle": ok}


def sources_sha16():
    """sha256 prefix of EVERY source file that reaches libbhw.so (blackman_harris_win_amd/_build.py::library_sources: the kernel units,
    the shared headers incl. bhw_tables.inc, the planner, the API layer, the Taylor ROM generator, include/bhw.h): ties a committed PMC
    summary to the code it was measured on -- a change in any of them marks the citation stale."""
    from blackman_harris_win_amd import _build
    h = hashlib.sha256()
    for path in sorted(_build.library_sources()):
        with open(path, "rb") as fh:
            h.update(os.path.basename(path).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def read_sclk(index=0):
    """Highest current shader clock (MHz) over the GPUs sysfs lists (pp_dpm_sclk: the level marked '*'), or None where nothing is
    readable.  A box may list more cards than HIP shows this process and in another order, so the rank's own card is not picked by
    index: the card running this benchmark's load is the one at the top.  Clock evidence that does not depend on the driver's coarse
    gpu_busy sampling."""
    import glob
    best = None
    for path in glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"):
        try:
            with open(path) as f:
                for line in f:
                    if line.rstrip().endswith("*"):
                        mhz = int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
                        best = mhz if best is None or mhz > best else best
        except Exception:
            continue
    return best


def pmc_traffic():
    """(bytes, source): HBM bytes per step from the committed rocprofv3 PMC passes of this same command (tools/gpu_pmc.sh ->
    profiles/pmc_latest.json): 2 x FETCH_SIZE + WRITE_SIZE per kernel, summed over the kernels of one step (the gfx950
    correction of MI355X_MICROARCH.md section HBM; WRITE_SIZE calibrated on the build kernel's known bytes).  A citation of
    an earlier profiling run, not a measurement of this run: the source record says which file, when, and on which source
    hash; a summary taken on other sources is not quoted (bytes = None)."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except Exception:
        return None, {"file": None, "note": "no committed PMC summary"}
    meta = d.get("_meta", {})
    src = {"file": "profiles/pmc_latest.json", "measured": meta.get("date"), "sources_sha16": meta.get("sources_sha16"),
           "command": meta.get("command"), "kind": "citation of a separate rocprofv3 --pmc run (tools/gpu_pmc.sh), not measured by this run"}
    if meta.get("sources_sha16") != sources_sha16():
        src["note"] = "stale: kernels changed since that PMC run (current sources_sha16 %s); traffic withheld" % sources_sha16()
        return None, src
    return float(d["_step_hbm_bytes"]), src


def pmc_legs_traffic():
    """{leg: {...}} from the committed per-leg counter passes (tools/gpu_pmc_legs.sh -> profiles/pmc_legs_latest.json): bytes written /
    fetched per call of every extra leg, cited (not measured by this run) and only when taken on the current kernel sources."""
    path = os.path.join(ROOT, "profiles", "pmc_legs_latest.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except Exception:
        return {}, {"file": None, "note": "no committed per-leg PMC summary"}
    meta = d.get("_meta", {})
    src = {"file": "profiles/pmc_legs_latest.json", "measured": meta.get("date"), "sources_sha16": meta.get("sources_sha16"),
           "kind": "citation of separate rocprofv3 --pmc runs (tools/gpu_pmc_legs.sh), not measured by this run"}
    if meta.get("sources_sha16") != sources_sha16():
        src["note"] = "stale: kernels changed since those PMC runs (current sources_sha16 %s); traffic withheld" % sources_sha16()
        return {}, src
    return {k: v for k, v in d.items() if not k.startswith("_")}, src


def self_launch(n_ranks, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE: run the N ranks as children of this process (torch.distributed.run on
    127.0.0.1) and return their exit code.  Called before anything in this process has initialised a GPU; the ranks are new
    processes, nothing is re-exec'ed."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")         # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def launch_check(args, rank, world):
    """--launch-check: the N>1 plumbing without a GPU (tests/test_distributed.py): process group over gloo, the timing protocol
    on a sleeping step, rank 0 prints the JSON line."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group(backend="gloo")

    def allreduce_max(v):
        if world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    t_own = time.perf_counter()
    elapsed = timed_steps(lambda: time.sleep(0.002 * (rank + 1)), args.steps, args.warmup,
                          dist.barrier if world > 1 else (lambda: None), lambda: None, allreduce_max)
    own_ms = 0.002 * (rank + 1) * 1e3                       # this rank's own step time (stands in for the device time per step)
    del t_own
    group = group_record(dist if world > 1 else None, torch, "cpu", world, "gloo", own_ms)
    strong_rec = None
    if world > 1 and args.scaling == "weak":                    # like the GPU run: a weak multi-rank record also times the strong reading
        s_el = timed_steps(lambda: time.sleep(0.002 * (rank + 1) / world), args.steps, args.warmup, dist.barrier, lambda: None, allreduce_max)
        strong_rec = {"value": COUNT * args.steps / s_el / 1e9, "unit": "Gsamples/s", "ms_per_step": s_el / args.steps * 1e3,
                      "steps": args.steps, "parity_spot_check": None, "launch_check": True}
    if rank == 0:
        print(json.dumps(dict({"launch_check": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": args.scaling,
                               "ms_per_step": elapsed / args.steps * 1e3, "shard_of_last_rank": list(shard_for(world - 1)),
                               "strong": strong_rec}, **group)), flush=True)
    if world > 1:
        dist.destroy_process_group()


def group_record(dist, torch, device, world, backend, own_device_ms):
    """What the process group itself reports, so a scaling record can be checked against it: `ranks_seen` = all-reduce SUM of 1
    over the group (must equal n_gpus), `device_ms_per_step_by_rank` = every rank's own device time per step (all-gather), and the
    backend name.  Collectives on the timing path only -- the data path has none."""
    if dist is None:
        return {"ranks_seen": 1, "device_ms_per_step_by_rank": [own_device_ms], "backend": None}
    one = torch.ones(1, dtype=torch.float64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    mine = torch.tensor([own_device_ms], dtype=torch.float64, device=device)
    every = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
    dist.all_gather(every, mine)
    return {"ranks_seen": int(round(float(one.item()))), "device_ms_per_step_by_rank": [float(t.item()) for t in every],
            "backend": backend + (" (RCCL)" if backend == "nccl" else "")}


def device_times(step, steps, torch):
    """Per-step device time (ms) of `steps` back-to-back steps: one HIP event before each step and one after the last, on
    the stream the kernels are launched on.  Outside the timed region (the extra events would perturb it)."""
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    for i in range(steps):
        evs[i].record()
        step()
    evs[steps].record()
    torch.cuda.synchronize()
    return [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]


def golden_parity_owned(torch, out, segments, dev):
    """Every committed golden sample of the C3 window that lies inside `segments` (this rank's ownership) matches: 0.0 yes, 1.0 no,
    2.0 the fixture is missing.  (A float so that the ranks can max-reduce it.)"""
    step_s = (1 << 23) // 1024
    try:
        with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
            g = json.load(f)["entries"]["C3_bh7_26_32"]
        owned = torch.zeros(COUNT, dtype=torch.bool, device=dev)
        for s0, c in segments:
            owned[s0:s0 + c] = True
        bad = 0.0
        for sh, e in enumerate(g["shards"]):
            idx = (sh << 23) + step_s * torch.arange(1024, device=dev)
            want = torch.tensor(e["strided_1024"], dtype=torch.int32, device=dev)
            m = owned[idx]
            if not bool((out[idx][m] == want[m]).all()):
                bad = 1.0
        return bad
    except Exception:
        return 2.0


def strong_leg(torch, bhw, B, params, rank, world, dev, algo, steps, warmup, barrier, allreduce_max):
    """BASELINE configs[4] as SURVEY 8(d) defines it -- ONE 2^26-point window over the ranks -- timed with the same protocol as the
    headline, so that a multi-rank record of the default (weak) command carries both readings of "8 GPUs": `strong` = interleaved
    ownership parts of one window (bhw_generate_part_device, no collective).  Returns the dict rank 0 prints as rec["strong"]."""
    out = torch.empty(COUNT, dtype=torch.int32, device=dev)
    ws_bytes = B.lib().bhw_workspace_bytes(ctypes.byref(params), 0, COUNT, B.ALGO_TABLE)
    workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    segments = B.part_segments(params, rank, world)

    def step():
        bhw.generate_part(params, rank, world, out, algo=algo, workspace=workspace)

    step()
    torch.cuda.synchronize()
    elapsed = timed_steps(step, steps, warmup, barrier, torch.cuda.synchronize, allreduce_max)
    worst = allreduce_max(golden_parity_owned(torch, out, segments, dev))
    return {"value": COUNT * steps / elapsed / 1e9, "unit": "Gsamples/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps,
            "parity_spot_check": None if worst == 2.0 else worst == 0.0,
            "workload": "ONE BH-7 2^26 / 32-bit window over the %d ranks: interleaved ownership parts, no collective (SURVEY 8(d) C5)" % world,
            "coefficients_of_rank0": sum(c for _, c in segments),
            "note": "same timing protocol as the headline (barrier + synchronize around K steps, max over ranks); value = 2^26 x K / time"}


def extra_legs(torch, bhw, B, out, steps):
    """Untimed legs for the other BASELINE configurations and for the variants DESIGN.md quotes, each measured like the headline's
    device time: a short ramp, then the median of per-step HIP-event intervals on the launch stream (short windows: 20 calls
    captured into one HIP graph, per-window time = replay time / 20, so the host's ~7 us per call is not in the figure).
    {ms, GB/s, frac, plan} per leg; bytes = 4 B per coefficient written (fused apply: 8 B, x read + y written)."""
    legs = {}

    def measure(fn, n_coeff, plan, bytes_per=BYTES_PER_COEFF, note=None, reps=None):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:                  # the previous leg's read-backs let the clocks drop
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        reps = reps or min(max(steps, 10), 50)
        batches = []
        for _ in range(5):                                      # like the headline's device time: events around back-to-back steps
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            batches.append(e0.elapsed_time(e1) / reps)
        ms = statistics.median(batches)
        gbs = bytes_per * n_coeff / (ms * 1e-3) / 1e9
        leg = {"ms": ms, "GB/s": gbs, "frac": gbs / HBM_PEAK_GBS, "Gsamples_per_s": n_coeff / (ms * 1e-3) / 1e9,
               "plan": plan() if callable(plan) else plan}
        if note:
            leg["note"] = note
        return leg

    n26 = 1 << 26
    # BASELINE configs[1]: BH-4, N = 2^20, 24-bit -- per call, and per window inside a 20-call HIP graph
    p2 = bhw.make_params(4, 20, 24)
    bhw.prepare(p2)
    o2 = out[:1 << 20]
    plan2 = B.describe_plan(p2, 0, 1 << 20, B.ALGO_AUTO)
    legs["C2_bh4_2^20_24bit_per_call"] = measure(lambda: bhw.generate(p2, 0, 1 << 20, out=o2), 1 << 20, plan2,
                                                 note="back-to-back calls from Python: includes the host's per-call cost", reps=200)
    try:
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            bhw.generate(p2, 0, 1 << 20, out=o2)
            st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for _ in range(20):
                    bhw.generate(p2, 0, 1 << 20, out=o2)
            leg = measure(g.replay, 20 << 20, plan2, note="20 windows per HIP-graph replay; ms = per window", reps=50)
        leg["ms"] /= 20.0
        legs["C2_bh4_2^20_24bit_graph"] = leg
    except Exception as e:                                       # graph capture unavailable: the per-call figure stands
        legs["C2_bh4_2^20_24bit_graph"] = {"error": repr(e)}
    # the same 20 windows as THROUGHPUT instead of latency: 20 different buffers, four streams of five calls inside one graph (the
    # streaming-frame use of configs[3]: independent windows, nothing orders one behind the other but its stream)
    try:
        bufs = [torch.empty(1 << 20, dtype=torch.int32, device=out.device) for _ in range(20)]
        main_st = torch.cuda.Stream()
        side = [torch.cuda.Stream() for _ in range(4)]
        for s_ in side:                                          # every stream gets its scratch / lazy state before the capture
            with torch.cuda.stream(s_):
                bhw.prepare(p2)
                bhw.generate(p2, 0, 1 << 20, out=bufs[0])
        torch.cuda.synchronize()
        with torch.cuda.stream(main_st):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=main_st):
                fork = torch.cuda.Event()
                fork.record(main_st)
                for si, s_ in enumerate(side):
                    s_.wait_event(fork)
                    with torch.cuda.stream(s_):
                        for k in range(5):
                            bhw.generate(p2, 0, 1 << 20, out=bufs[si * 5 + k])
                    join = torch.cuda.Event()
                    join.record(s_)
                    main_st.wait_event(join)
            leg = measure(g.replay, 20 << 20, plan2, note="20 windows into 20 buffers over 4 streams per HIP-graph replay; ms = per window "
                          "(throughput: windows overlap); the dependent-chain figure is C2_bh4_2^20_24bit_graph", reps=50)
        leg["ms"] /= 20.0
        leg["windows_per_s"] = 1.0 / (leg["ms"] * 1e-3)
        legs["C2_bh4_2^20_24bit_graph_4streams"] = leg
        del bufs
    except Exception as e:
        legs["C2_bh4_2^20_24bit_graph_4streams"] = {"error": repr(e)}
    # the streaming-frame size of configs[3] with the headline's window type: BH-7, N = 2^16, 32-bit, 20 windows per graph replay
    p7s = bhw.make_params(WIN, 16, DAT_WIDTH)
    bhw.prepare(p7s)
    o7s = out[:1 << 16]
    plan7s = B.describe_plan(p7s, 0, 1 << 16, B.ALGO_AUTO)
    try:
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            bhw.generate(p7s, 0, 1 << 16, out=o7s)
            st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for _ in range(20):
                    bhw.generate(p7s, 0, 1 << 16, out=o7s)
            leg = measure(g.replay, 20 << 16, plan7s, note="20 windows per HIP-graph replay; ms = per window", reps=50)
        leg["ms"] /= 20.0
        legs["bh7_2^16_32bit_graph"] = leg
    except Exception as e:
        legs["bh7_2^16_32bit_graph"] = {"error": repr(e)}
    # BASELINE configs[3]: 1024 frames x BH-4 N = 2^16, 24-bit (one period computed, then store-only replication)
    p4 = bhw.make_params(4, 16, 24)
    o4 = out.view(1024, 1 << 16)
    legs["C4_1024x_bh4_2^16_24bit"] = measure(lambda: bhw.generate_batched(p4, 1024, out=o4), n26,
                                              lambda: B.describe_plan(p4, 0, 1 << 16, B.ALGO_AUTO) + " -- batched: the one launch writes all 1024 frames")
    # phase bits dropped (models A / C at PHASE_WIDTH > DATA_WIDTH): run-length kernel
    pn = bhw.make_params(WIN, PHI_WIDTH, 16, model=B.MODEL_CPP)
    legs["bh7_2^26_16bit_cpp"] = measure(lambda: bhw.generate(pn, 0, n26, out=out), n26, lambda: B.describe_plan(pn, 0, n26, B.ALGO_AUTO))
    # Taylor source (win_selector wires it to Hamming / BH-3 only)
    pt = bhw.make_params(1, PHI_WIDTH, 16, sin_type=B.SIN_TAYLOR, combine=B.COMBINE_VHDL, lut_size=9)
    legs["taylor_hamming_2^26_16bit"] = measure(lambda: bhw.generate(pt, 0, n26, out=out), n26, lambda: B.describe_plan(pt, 0, n26, B.ALGO_AUTO))
    # the headline window with the VHDL cosine-sum (src/bh_win_7term.vhd:353-438: what win_selector instantiates)
    pv = bhw.make_params(WIN, PHI_WIDTH, DAT_WIDTH, combine=B.COMBINE_VHDL)
    legs["C3_vhdl_cosine_sum"] = measure(lambda: bhw.generate(pv, 0, n26, out=out), n26, lambda: B.describe_plan(pv, 0, n26, B.ALGO_AUTO))
    # ... and with the VHDL CORDIC as well (model C end to end)
    pvv = bhw.make_params(WIN, PHI_WIDTH, DAT_WIDTH, model=B.MODEL_VHDL, combine=B.COMBINE_VHDL)
    legs["C3_vhdl_cordic_and_sum"] = measure(lambda: bhw.generate(pvv, 0, n26, out=out), n26, lambda: B.describe_plan(pvv, 0, n26, B.ALGO_AUTO))
    # Pipelined THROUGHPUT of the headline window -- never the headline: successive windows alternate between two streams (own output
    # buffer and own library scratch each), so the build pass of window i + 1 (vector-issue bound, writes 22 MB) can run beside the
    # combine pass of window i (memory bound).  Every window still builds its own table.
    try:
        p3s = bhw.make_params(WIN, PHI_WIDTH, DAT_WIDTH)
        out_b = torch.empty_like(out)
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        for s_ in (sa, sb):
            with torch.cuda.stream(s_):
                bhw.prepare(p3s)
        torch.cuda.synchronize()
        pairs = 25

        def two_streams():
            fork = torch.cuda.Event()
            fork.record()
            sa.wait_event(fork)
            sb.wait_event(fork)
            for _ in range(pairs):
                with torch.cuda.stream(sa):
                    bhw.generate(p3s, 0, n26, out=out)
                with torch.cuda.stream(sb):
                    bhw.generate(p3s, 0, n26, out=out_b)
            cur = torch.cuda.current_stream()
            for s_ in (sa, sb):
                join = torch.cuda.Event()
                join.record(s_)
                cur.wait_event(join)

        leg = measure(two_streams, 2 * pairs * n26, lambda: B.describe_plan(p3s, 0, n26, B.ALGO_AUTO) + " -- windows alternating over two streams",
                      note="throughput of back-to-back windows on two streams (build of one beside the combine of the other); ms = per window; "
                           "the latency of ONE window is the headline", reps=4)
        leg["ms"] /= 2 * pairs
        leg["same_as_headline"] = bool(torch.equal(out, out_b))
        legs["C3_two_streams"] = leg
        del out_b
    except Exception as e:
        legs["C3_two_streams"] = {"error": repr(e)}
    # fused apply y = (x * w) >> 31: reads x, writes y, no coefficient vector in HBM
    x = torch.randint(-(1 << 31), (1 << 31) - 1, (n26,), dtype=torch.int32, device=out.device)
    p3 = bhw.make_params(WIN, PHI_WIDTH, DAT_WIDTH)
    legs["fused_apply_C3"] = measure(lambda: bhw.apply(p3, x, out=out, shift=31), n26, lambda: "bhw_apply_device: " + B.describe_plan(p3, 0, n26, B.ALGO_AUTO),
                                     bytes_per=8, note="8 B per sample: x read + y written")
    del x
    return legs


def spread(ms):
    srt = sorted(ms)
    return {"min": srt[0], "median": statistics.median(srt), "p90": srt[min(len(srt) - 1, int(0.9 * len(srt)))], "max": srt[-1],
            "n": len(srt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every rank generates a full 2^26 window; strong: ONE 2^26 window over the ranks "
                         "(interleaved ownership parts, BASELINE config C5)")
    ap.add_argument("--algo", default="auto", choices=["auto", "direct", "table", "fused"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpp-leg", action="store_true", help="skip the extra model-cpp leg (profiling passes: only the headline kernels run)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--ramp-seconds", type=float, default=1.0, help="untimed back-to-back steps before the warmup (clock ramp)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for the barrier / max-over-ranks (nccl = RCCL; gloo only to rehearse N>1 "
                         "on a box with fewer GPUs than ranks, together with BHW_BENCH_SHARE_GPU=1)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the untimed legs for the other BASELINE configurations")
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))           # this process has not touched a GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         "(or leave WORLD_SIZE unset: bench.py then starts its ranks itself)")
    if args.launch_check:
        return launch_check(args, rank, world)

    import torch
    import blackman_harris_win_amd as bhw
    from blackman_harris_win_amd import binding as B

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the generator has no CPU path)")
    if os.environ.get("BHW_BENCH_SHARE_GPU") == "1":        # rehearsal only: all ranks on one GPU
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)   # "nccl" is RCCL on ROCm; used for barrier/max only
        else:
            dist.init_process_group(backend="gloo")

    algo = {"auto": B.ALGO_AUTO, "direct": B.ALGO_DIRECT, "table": B.ALGO_TABLE, "fused": B.ALGO_FUSED}[args.algo]
    params = bhw.make_params(WIN, PHI_WIDTH, DAT_WIDTH)          # model HLS + HLS combine + built-in a_k
    strong = args.scaling == "strong"
    workspace = None
    if strong:
        # one window over the ranks: this rank's ownership part, written in place into a full-length buffer
        out = torch.empty(COUNT, dtype=torch.int32, device=dev)
        ws_bytes = B.lib().bhw_workspace_bytes(ctypes.byref(params), 0, COUNT, B.ALGO_TABLE)
        workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        segments = B.part_segments(params, rank, world)
        units_per_step = COUNT                                 # whole job: one window per step
        my_units = sum(c for _, c in segments)
        plan = "interleaved ownership part %d/%d (%d segments, %d coefficients)" % (rank, world, len(segments), my_units)

        def step():
            bhw.generate_part(params, rank, world, out, algo=algo, workspace=workspace)
    else:
        n0, count = shard_for(rank)
        out = torch.empty(count, dtype=torch.int32, device=dev)
        # scratch: sized for the table format this configuration uses once bhw_prepare_device has settled it (16.5 MiB of nibble
        # entries + records for the headline window; the format-independent bound bhw_workspace_bytes is 128 MiB)
        bhw.prepare(params)
        ex_q = B.BhwExec()
        ex_q.struct_size = ctypes.sizeof(B.BhwExec)
        ex_q.algo = algo
        ws_bytes = B.lib().bhw_workspace_bytes_ex(ctypes.byref(params), n0, count, ctypes.byref(ex_q))
        workspace = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
        units_per_step = world * count
        my_units = count

        def step():
            bhw.generate(params, n0, count, out=out, algo=algo, workspace=workspace if ws_bytes else None)

    def barrier():
        if dist is not None:
            dist.barrier()

    def allreduce_max(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # device-side duration of the K timed steps on the launch stream (HIP events), for the roofline figure
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    state = {"n": 0}

    def step_with_events():
        if state["n"] == 0:
            ev0.record()
        step()
        state["n"] += 1
        if state["n"] == args.steps:
            ev1.record()

    # untimed: settle the lazy per-configuration work (table format verification) and let the device clocks ramp -- the first
    # steps after idle run up to ~10 % slower, and a cold box needs about a second of load before the rate is flat
    step()
    torch.cuda.synchronize()
    # the same K steps straight after W warm-up steps, BEFORE the ramp: what a cold device gives (DVFS), printed beside the
    # ramped figure so the effect of the ramp is in the record and not in prose
    no_ramp_elapsed = timed_steps(step, args.steps, args.warmup, barrier, torch.cuda.synchronize, allreduce_max)
    t_ramp = time.perf_counter()
    ramp_steps = 0
    sclk_before = None
    while time.perf_counter() - t_ramp < args.ramp_seconds:
        for _ in range(50):
            step()
        ramp_steps += 50
        sclk_before = read_sclk(local_rank)        # sampled while the 50 steps just enqueued are running: the clock under this load
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    ev0.record()                                   # torch creates the HIP event handle at the first record(): not inside the timed region
    ev1.record()                                   # (two hipEventCreate calls cost ~25 us there -- 1 % of a 20-step run)
    elapsed = timed_steps(step_with_events, args.steps, 0, barrier, torch.cuda.synchronize, allreduce_max)
    for _ in range(50):                            # (outside the timed region) the same load again, sampled while it runs
        step()
    sclk_after = read_sclk(local_rank)
    torch.cuda.synchronize()
    own_dev_ms = ev0.elapsed_time(ev1) / args.steps
    dev_ms = allreduce_max(own_dev_ms)
    group = group_record(dist, torch, dev if args.backend == "nccl" else "cpu", world, args.backend, own_dev_ms)

    # per-step device-time distribution over K more steps (outside the timed region)
    step_ms = spread(device_times(step, args.steps, torch))

    # per-kernel split, outside the timed region: a few extra steps with the library's event recorded between
    # the table build and the combine pass (cross-check for the rocprofv3 kernel stats under profiles/)
    plan_line = B.describe_plan(params, 0 if strong else n0, COUNT, algo) if not strong else None
    per_kernel = None
    if not strong and plan_line.startswith("table"):
        names = plan_line.split(": ", 1)[1].split(" + ")
        reps = min(10, args.steps)
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(reps)]
        for trio in evs:
            for e in trio:
                e.record()                 # creates the handles; re-recorded below
        torch.cuda.synchronize()
        for e0, em, e1 in evs:
            e0.record()
            bhw.generate(params, n0, count, out=out, algo=algo, workspace=workspace, event_after_build=em)
            e1.record()
        torch.cuda.synchronize()
        build_us = sum(t[0].elapsed_time(t[1]) for t in evs) / reps * 1e3
        comb_us = sum(t[1].elapsed_time(t[2]) for t in evs) / reps * 1e3
        per_kernel = {names[0]: {"avg_us": build_us},
                      names[1].split(" ")[0]: {"avg_us": comb_us, "dominant": True,
                                               "achieved_alone_GBps": BYTES_PER_COEFF * count / (comb_us * 1e-6) / 1e9},
                      "note": "event-to-event, includes the launch gap; rocprofv3 kernel stats under profiles/"}

    # parity check outside the timed region, against the committed golden fixture of this exact config
    # (tests/golden/golden.json, C3: sparse samples, per-shard sums and 8 x 1024 strided samples); the full
    # bit-for-bit comparison with the oracle is tests/test_gpu_parity.py
    parity = None
    step_s = (1 << 23) // 1024
    if strong and world > 1:
        # this rank holds only its segments: every golden sample inside them must match.  Every rank reaches the collective
        # whatever happened to it (a mismatch or a missing fixture on one rank must not leave the others waiting in it).
        worst = allreduce_max(golden_parity_owned(torch, out, segments, dev))
        parity = None if worst == 2.0 else worst == 0.0
    try:
        with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
            g = json.load(f)["entries"]["C3_bh7_26_32"]
        if strong and world > 1:
            pass
        else:
            parity = all(int(out[int(n)]) == v for n, v in g["sparse"].items())
            for sh, e in enumerate(g["shards"]):
                seg = out[sh << 23:(sh + 1) << 23]
                parity = parity and int(seg.sum(dtype=torch.int64)) == e["sum"]
                parity = parity and seg[::step_s][:1024].cpu().tolist() == e["strided_1024"]
            parity = bool(parity)
    except FileNotFoundError:
        pass

    # the same step with the cpp model's cosines (cpp/cordic_sincos.cpp: the bit-model pinned by the reference's own compiled
    # cordic(), tests/test_gpu_reference_pin.py) -- untimed extra leg, so the pinned model's rate is in the record too
    cpp_leg = None
    if not strong and rank == 0 and not args.no_cpp_leg:
        pc = bhw.make_params(WIN, PHI_WIDTH, DAT_WIDTH, model=B.MODEL_CPP)

        def step_cpp():
            bhw.generate(pc, n0, count, out=out, algo=algo)      # library scratch: this model's table format is wider than the headline's
        step_cpp()                                              # settles this configuration's table format
        torch.cuda.synchronize()
        t_r = time.perf_counter()                               # the parity check above let the clocks drop: ramp again
        while time.perf_counter() - t_r < 0.3:
            for _ in range(50):
                step_cpp()
            torch.cuda.synchronize()
        ms = spread(device_times(step_cpp, min(args.steps, 50), torch))
        cpp_leg = {"ms_per_step_median": ms["median"], "Gsamples_per_s": count / (ms["median"] * 1e-3) / 1e9,
                   "plan": B.describe_plan(pc, n0, count, algo),
                   "note": "same window, CORDIC bit-model of cpp/cordic_sincos.cpp (BHW_MODEL_CPP) in the HLS cosine-sum; device time"}
        step()                                                  # leave the headline window in `out`

    # what the library itself holds for this stream (the cpp leg and the extra legs run on it; the headline passes its own workspace)
    try:
        Lq = B.lib()
        Lq.bhw_dbg_library_scratch_bytes.restype = ctypes.c_uint64
        Lq.bhw_dbg_library_scratch_bytes.argtypes = [ctypes.c_int, ctypes.c_void_p]
        lib_scratch = int(Lq.bhw_dbg_library_scratch_bytes(local_rank, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    except Exception:
        lib_scratch = None

    legs = None
    if not strong and rank == 0 and world == 1 and not args.no_extra_legs:
        legs = extra_legs(torch, bhw, B, out, args.steps)
        if cpp_leg:
            gbs = BYTES_PER_COEFF * count / (cpp_leg["ms_per_step_median"] * 1e-3) / 1e9
            legs["C3_model_cpp"] = {"ms": cpp_leg["ms_per_step_median"], "GB/s": gbs, "frac": gbs / HBM_PEAK_GBS,
                                    "Gsamples_per_s": cpp_leg["Gsamples_per_s"], "plan": cpp_leg["plan"]}
        # counted bytes beside every leg (profiles/pmc_legs_latest.json, hash-tied to the kernel sources)
        cited, cited_src = pmc_legs_traffic()
        alias = {"C2_bh4_2^20_24bit_per_call": "C2_bh4_2^20_24bit", "C2_bh4_2^20_24bit_graph": "C2_bh4_2^20_24bit",
                 "C2_bh4_2^20_24bit_graph_4streams": "C2_bh4_2^20_24bit", "C3_two_streams": "headline_C3", "bh7_2^16_32bit_graph": "bh7_2^16_32bit"}
        for name, leg in legs.items():
            c = cited.get(alias.get(name, name))
            if isinstance(leg, dict) and "ms" in leg:
                leg["traffic"] = None if c is None else {"bytes_per_call": c["traffic_bytes_per_call"], "write_bytes": c["write_bytes_per_call"],
                                                         "read_bytes": c["read_bytes_per_call"], "over_algorithmic": c["traffic_over_algorithmic"]}
                if c is not None and leg["ms"] > 0:
                    leg["counted_write_GB/s"] = c["write_bytes_per_call"] / (leg["ms"] * 1e-3) / 1e9
        legs["_traffic_source"] = cited_src
        step()

    # A multi-rank record of the default (weak) command also carries the strong reading of configs[4] -- one window over the ranks --
    # so that one SCALE record tells the whole story: weak scales trivially, strong is bounded by the table each part rebuilds
    strong_rec = None
    if world > 1 and not strong:
        strong_rec = strong_leg(torch, bhw, B, params, rank, world, dev, algo, args.steps, args.warmup, barrier, allreduce_max)

    total = units_per_step * args.steps
    value = total / elapsed / 1e9
    # roofline: algorithmic bytes of what THIS device wrote per step / its device time per step
    achieved = BYTES_PER_COEFF * my_units / (dev_ms * 1e-3) / 1e9
    traffic, traffic_source = pmc_traffic() if (not strong and args.algo in ("auto", "table")) else (None, {"file": None})
    rec = {
        "metric": "window Gsamples/s (BH-7, N=2^26, 32-bit) + fraction of HBM-write roofline",
        "value": value, "unit": "Gsamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "int64", "output_dtype": "int32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: Blackman-Harris 7-term, N=2^26 (64M), 32-bit output, model HLS CORDIC + HLS cosine-sum"
                               + (" -- ONE window over the ranks (configs[4], C5)" if strong else ""),
                   "phi_width": PHI_WIDTH, "dat_width": DAT_WIDTH, "n_terms": 7,
                   "coefficients_per_step_per_gpu": my_units, "strategy": args.algo,
                   "plan": plan if strong else plan_line,
                   "sharding": ("interleaved ownership parts of one window, no collective" if strong else
                                "contiguous stream-index range per rank (one full window each), no collective")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_source,
                     "device_ms_per_step": dev_ms, "device_ms_per_step_spread": step_ms,
                     "algorithmic_bytes": BYTES_PER_COEFF * my_units, "per_kernel": per_kernel,
                     "note": "achieved = 4 B x coefficients this device writes per step / device time of one step (every kernel of the "
                             "step; HIP events on the launch stream around the K timed steps); spread = per-step event times of K "
                             "further steps; dtype = widest arithmetic type on the path (64-bit CORDIC state), output int32"},
        "ramp": {"seconds": args.ramp_seconds, "steps": ramp_steps,
                 "no_ramp": {"ms_per_step": no_ramp_elapsed / args.steps * 1e3, "value": total / no_ramp_elapsed / 1e9,
                             "note": "the same W warm-up + K timed steps run BEFORE the ramp (cold clocks); `value` above is after it"}},
        "ranks_seen": group["ranks_seen"], "device_ms_per_step_by_rank": group["device_ms_per_step_by_rank"], "backend": group["backend"],
        "scratch_bytes": int(workspace.numel()) if workspace is not None else 0,
        "library_scratch_bytes": lib_scratch,
        "sclk_mhz": {"end_of_ramp": sclk_before, "after_timed_region": sclk_after,
                     "source": "sysfs pp_dpm_sclk (highest over the cards listed), read by the host while 50 enqueued steps are running -- at the end of "
                               "the ramp (just before the W warm-up + K timed steps) and right after the timed region (None: not readable on "
                               "this box; an idle device reads ~100 MHz)"},
        "strong": strong_rec,
        "parity_spot_check": parity,
        "cpp_model": cpp_leg,
        "extra_legs": legs,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        rec["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.cpu_threads)
    elif rank == 0:
        rec["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
