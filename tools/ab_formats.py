#!/usr/bin/env python3
"""Table formats of the headline window timed side by side (event-timed Python loops, interleaved rounds)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import blackman_harris_win_amd as bhw
from blackman_harris_win_amd import binding as B

for win, pw, w, model in ((7, 26, 32, 0), (7, 26, 32, 1), (7, 24, 30, 0)):
    p = bhw.make_params(win, pw, w, model=model)
    n = 1 << pw
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    fmts = [("nibble", B.TABLE_NIBBLE), ("residual", B.TABLE_RESIDUAL), ("delta16", B.TABLE_DELTA16), ("plain", B.TABLE_PLAIN)]
    for _ in range(300):
        bhw.generate(p, 0, n, out=out)
    torch.cuda.synchronize()
    res = {k: [] for k, _ in fmts}
    for r in range(6):
        for name, f in fmts:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                bhw.generate(p, 0, n, out=out, table_format=f)
            e1.record()
            torch.cuda.synchronize()
            if r:
                res[name].append(e0.elapsed_time(e1) / 100)
    print((win, pw, w, model), {k: round(statistics.median(v), 4) for k, v in res.items()})
