"""The N>1 path on CPU (gloo, world_size 2): bench.py's shard plan, barrier/max-over-ranks protocol, the index-range sharding
of the selector and the interleaved ownership parts (bhw_part_segments), with the oracle standing in for the GPU step."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNOWNED = -(1 << 40)          # outside the int32 range: coefficients may be negative (caller-scaled weights, Hann at 10/24)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import time
    import torch
    import torch.distributed as dist
    import bench
    import oracle_lib as O
    from blackman_harris_win_amd import shard_range

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    # (1) strong-scaling shard of one window (SURVEY 8e / config C5 at small size): contiguous, no collective
    p = O.oparams(7, 12, 32)
    n0, count = shard_range(1 << 12, rank, world)
    shard = O.generate(p, n0, count)
    np.save(os.path.join(outdir, f"shard{rank}.npy"), shard)
    # (1b) interleaved ownership of one window (bench.py --scaling strong, bhw_generate_part_device): every rank fills exactly
    # the segments bhw_part_segments gives it -- host arithmetic of the C ABI, no GPU -- and nothing else
    from blackman_harris_win_amd import binding as B
    for win, pw, w in ((7, 22, 32), (4, 16, 24)):
        bp = B.make_params(win, pw, w)
        po = O.from_bhw(bp)
        window = np.full(1 << pw, UNOWNED, np.int64)
        for n0s, cnt in B.part_segments(bp, rank, world):
            window[n0s:n0s + cnt] = O.generate_mt(po, n0s, cnt, threads=2)
        np.save(os.path.join(outdir, f"part_{win}_{pw}_{rank}.npy"), window)
    # (2) bench.py's weak-scaling plan: rank r owns stream indices [r*2^26, (r+1)*2^26)
    b0, bc = bench.shard_for(rank)
    assert (b0, bc) == (rank << 26, 1 << 26)
    # (3) the timing protocol: barrier + sync brackets, MAX over ranks
    calls = {"n": 0}

    def step():
        calls["n"] += 1
        time.sleep(0.01 * (rank + 1))          # rank 1 is slower: the reported time must be its time

    def allreduce_max(v):
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed = bench.timed_steps(step, steps=5, warmup=2, barrier=dist.barrier, sync=lambda: None, allreduce_max=allreduce_max)
    assert calls["n"] == 7
    assert elapsed >= 5 * 0.01 * world * 0.9
    gathered = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, torch.tensor([elapsed], dtype=torch.float64))
    assert all(abs(float(g) - elapsed) < 1e-12 for g in gathered)      # every rank reports the same max
    dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    whole = O.generate(O.oparams(7, 12, 32), 0, 1 << 12)
    got = np.concatenate([np.load(tmp_path / "shard0.npy"), np.load(tmp_path / "shard1.npy")])
    assert np.array_equal(got, whole)
    # the two ranks' interleaved parts tile the window: every coefficient owned (seam overlaps agree), values exact
    for win, pw, w in ((7, 22, 32), (4, 16, 24)):
        a, b = (np.load(tmp_path / f"part_{win}_{pw}_{r}.npy") for r in (0, 1))
        both = (a != UNOWNED) & (b != UNOWNED)
        assert ((a != UNOWNED) | (b != UNOWNED)).all() and both.sum() < (1 << pw) // 100
        assert np.array_equal(a[both], b[both])
        full = np.where(a != UNOWNED, a, b).astype(np.int32)
        assert np.array_equal(full, O.generate_mt(O.oparams(win, pw, w), 0, 1 << pw))


def _bench_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE (the form the driver uses at N = 1) starts its two ranks itself: the launch
    path, process group, barrier / max-over-ranks protocol and the one JSON line from rank 0, over gloo without a GPU."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--launch-check"],
                       capture_output=True, env=_bench_env(), timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                       # rank 0 only
    rec = json.loads(lines[0])
    assert rec["launch_check"] and rec["n_gpus"] == 2 and rec["steps"] == 5
    assert rec["ms_per_step"] >= 0.9 * 2 * 2.0                   # the slower rank's time (rank 1 sleeps 4 ms per step)
    assert rec["shard_of_last_rank"] == [1 << 26, 1 << 26]
    # the record checks itself: the process group saw both ranks, and every rank's own time is in it (rank r sleeps 2 (r + 1) ms)
    assert rec["ranks_seen"] == 2 and rec["backend"] == "gloo"
    assert rec["device_ms_per_step_by_rank"] == [2.0, 4.0]
    # a weak multi-rank record also carries the strong reading of configs[4] (ONE window over the ranks), timed the same way
    assert rec["scaling"] == "weak" and set(rec["strong"]) >= {"value", "ms_per_step", "parity_spot_check"}
    assert rec["strong"]["ms_per_step"] >= 0.9 * 2.0 and rec["strong"]["value"] > 0      # rank 1 sleeps 4 ms / 2 ranks per step


def test_bench_refuses_mismatched_world():
    """Under a launcher that set WORLD_SIZE the rank count must match --gpus (no second launch from inside a rank)."""
    import subprocess
    env = dict(_bench_env(), WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, env=env, timeout=300)
    assert r.returncode != 0 and b"WORLD_SIZE" in r.stderr


@pytest.mark.gpu
def test_bench_strong_scaling_two_ranks_share_one_gpu():
    """bench.py --gpus 2 --scaling strong end to end through its own launcher: one 2^26 window as two interleaved ownership parts,
    both ranks on this box's one GPU over gloo (a rehearsal of the launch path, not a measurement), parity checked on every rank."""
    import json
    import subprocess
    env = dict(_bench_env(), BHW_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", "strong", "--backend", "gloo",
                        "--steps", "5", "--warmup", "2", "--ramp-seconds", "0.2"], capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    rec = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["parity_spot_check"] is True
    assert rec["value"] > 0
    assert rec["ranks_seen"] == 2 and rec["backend"] == "gloo" and len(rec["device_ms_per_step_by_rank"]) == 2
    assert max(rec["device_ms_per_step_by_rank"]) == pytest.approx(rec["roofline"]["device_ms_per_step"])
    assert rec["ramp"]["no_ramp"]["ms_per_step"] > 0
    assert rec["strong"] is None                                 # (the strong run itself is the record)


@pytest.mark.gpu
def test_bench_weak_record_carries_the_strong_reading():
    """bench.py --gpus 2 (default, weak) through its own launcher, both ranks on this box's one GPU over gloo: the one JSON line holds
    the weak headline AND `strong` -- one window over the two ranks, parity checked on every rank's own segments."""
    import json
    import subprocess
    env = dict(_bench_env(), BHW_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "5", "--warmup", "2",
                        "--ramp-seconds", "0.2", "--no-cpp-leg"], capture_output=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    rec = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["parity_spot_check"] is True and rec["ranks_seen"] == 2
    s = rec["strong"]
    assert s["parity_spot_check"] is True and s["value"] > 0 and s["ms_per_step"] > 0 and s["steps"] == 5
    assert "ONE BH-7 2^26" in s["workload"]
