#!/usr/bin/env python3
"""Targeted parity campaign for the tile path (whole periods at 2^22 .. 2^26, every table format, every model and cosine-sum
rule, built-in and random weights, ownership parts): GPU vs the threaded oracle, bit for bit.
usage: fuzz_tile.py <seconds> [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle_lib as O
import blackman_harris_win_amd as bhw
from blackman_harris_win_amd import binding as B


def fuzz(budget=60.0, seed=1, max_cases=None, pws=(22, 22, 23, 23, 24, 24, 25, 26), verbose=False):
    """Runs until `budget` seconds or `max_cases` cases; returns (cases, coefficients, plans); raises AssertionError on a mismatch."""
    rng = np.random.default_rng(seed)
    t0 = time.time(); cases = 0; samples = 0; plans = {}; last = t0
    while (budget is None or time.time() - t0 < budget) and (max_cases is None or cases < max_cases):
        if verbose and time.time() - last > 60:                          # progress line (long silent runs are taken to be hung)
            last = time.time(); print("... %d cases, %d coefficients so far" % (cases, samples), flush=True)
        win = int(rng.choice([1, 2, 3, 4, 5, 7])); model = int(rng.integers(0, 3)); combine = int(rng.integers(0, 2))
        pw = int(rng.choice(pws)); w = int(rng.integers(8, 33))
        if model == B.MODEL_HLS and pw > w + 2:
            w = int(rng.integers(max(8, pw - 2), 33))
        prec = int(rng.integers(1, 4)) if model == B.MODEL_VHDL else 1
        aa = [int(v) for v in rng.integers(-(1 << (w - 1)), 1 << (w - 1), 7)] if rng.random() < 0.4 else None
        if aa is not None and rng.random() < 0.5:                       # small weights: the one-instruction products apply
            aa = [v >> 3 for v in aa]
        try:
            p = B.make_params(win, pw, w, model=model, combine=combine, precision=prec, aa=aa)
        except B.BhwError:
            continue
        n = 1 << pw
        fmt = int(rng.choice([B.TABLE_BEST, B.TABLE_BEST, B.TABLE_NIBBLE, B.TABLE_NIBBLE_ESC, B.TABLE_RESIDUAL, B.TABLE_DELTA16, B.TABLE_PLAIN]))
        if rng.random() < 0.2:                                          # a configuration class whose deviations fit four bits
            model, combine_keep = B.MODEL_HLS, combine
            w = int(rng.integers(max(28, pw + 4), 33)) if pw + 4 <= 32 else 32
            fmt = int(rng.choice([B.TABLE_BEST, B.TABLE_NIBBLE, B.TABLE_NIBBLE_ESC]))
            p = B.make_params(win, pw, w, model=model, combine=combine_keep, precision=1,
                              aa=None if aa is None else [max(-(1 << (w - 1)), min((1 << (w - 1)) - 1, v)) for v in aa])
        n0 = n * int(rng.integers(0, 3)) if rng.random() < 0.7 else int(rng.integers(0, 4 * n))
        count = n if rng.random() < 0.8 else n + int(rng.integers(1, 100000))
        desc = dict(win=win, pw=p.phi_width, w=p.dat_width, model=p.model, combine=p.combine, prec=p.precision, aa=aa, fmt=fmt, n0=n0, count=count)
        want = O.generate_mt(O.from_bhw(p), n0, count)
        if rng.random() < 0.25 and n0 % n == 0 and count == n:          # the same window assembled from ownership parts
            G = int(rng.choice([2, 3, 5, 8]))
            out = torch.full((n,), -(1 << 31), dtype=torch.int32, device="cuda")
            for g in range(G):
                bhw.generate_part(p, g, G, out)
            got = out.cpu().numpy(); desc["parts"] = G
        else:
            got = bhw.generate(p, n0, count, algo=B.ALGO_TABLE, table_format=fmt).cpu().numpy()
        if not np.array_equal(got, want):
            bad = int(np.flatnonzero(got != want)[0])
            raise AssertionError(dict(desc, first_bad=bad, got=int(got[bad]), want=int(want[bad]), n_bad=int((got != want).sum())))
        plan = B.describe_plan(p, n0, count, algo=B.ALGO_TABLE, table_format=fmt).split(":")[0].replace(", unverified", "")
        plans[plan] = plans.get(plan, 0) + 1
        cases += 1; samples += count
    return cases, samples, plans


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    cases, samples, plans = fuzz(budget, int(sys.argv[2]) if len(sys.argv) > 2 else 1, verbose=True)
    print("tile fuzz: %d cases, %d coefficients, all bit-exact; plans %s" % (cases, samples, plans))


if __name__ == "__main__":
    main()
