#!/usr/bin/env python3
"""Nibble + escapes table format: the same coefficients as the residual format over a set of configurations, and the formats
of the pinned models' headline windows timed side by side."""
import os, sys, statistics, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import blackman_harris_win_amd as bhw
from blackman_harris_win_amd import binding as B

L = B.lib()
L.bhw_dbg_table_format_verdict.argtypes = [ctypes.POINTER(B.BhwParams), ctypes.c_uint32, ctypes.c_int]
if os.environ.get('ESC_PARITY', '1') == '1':
  bad = 0
  for win, pw, w, model in ((7, 26, 32, 1), (7, 26, 32, 2), (7, 26, 32, 0), (7, 24, 32, 1), (4, 22, 24, 2), (5, 23, 29, 0), (7, 22, 28, 2), (3, 22, 30, 1),
                            (7, 22, 32, 1), (7, 25, 26, 1), (7, 24, 24, 2), (7, 26, 30, 1), (5, 25, 31, 1)):
      p = B.make_params(win, pw, w, model=model)
      a = bhw.generate(p, (1 << pw) - 12345, (1 << pw) + 12345 + 999, algo=B.ALGO_TABLE, table_format=B.TABLE_RESIDUAL)
      b = bhw.generate(p, (1 << pw) - 12345, (1 << pw) + 12345 + 999, algo=B.ALGO_TABLE, table_format=B.TABLE_NIBBLE_ESC)
      c = bhw.generate(p, (1 << pw) - 12345, (1 << pw) + 12345 + 999, algo=B.ALGO_TABLE, table_format=B.TABLE_BEST)
      ok = bool((a == b).all()) and bool((a == c).all())
      bad += not ok
      L = B.lib()
      L.bhw_dbg_table_format_verdict.argtypes = [ctypes.POINTER(B.BhwParams), ctypes.c_uint32, ctypes.c_int]
      dl = ctypes.c_uint32(0)
      L.bhw_dbg_table_format_info(ctypes.byref(p), ctypes.byref(dl), None)
      vd = [L.bhw_dbg_table_format_verdict(ctypes.byref(p), k + dl.value, 0) for k in (16, 48, 0)] if dl.value else None
      print("verdicts (nibble, nibble+esc, residual; 1 exact, 2 overflows):", vd, end="  ")
      print((win, pw, w, model), "ok" if ok else "MISMATCH %d" % int((a != b).sum()), B.describe_plan(p, 0, 1 << pw).split("\n")[0][:160], flush=True)
      del a, b, c
  print("mismatching configurations:", bad, flush=True)
  if bad:
      sys.exit(1)
for win, pw, w, model in ((7, 26, 32, 1), (7, 26, 32, 2), (7, 26, 32, 0)):
    p = bhw.make_params(win, pw, w, model=model)
    n = 1 << pw
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    fmts = [("best", B.TABLE_BEST), ("nibble+esc", B.TABLE_NIBBLE_ESC), ("residual", B.TABLE_RESIDUAL)]
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
    dl = ctypes.c_uint32(0)
    L.bhw_dbg_table_format_info(ctypes.byref(p), ctypes.byref(dl), None)
    vd = [L.bhw_dbg_table_format_verdict(ctypes.byref(p), k + dl.value, 0) for k in (16, 48, 0)]
    print((win, pw, w, model), {k: round(statistics.median(v), 4) for k, v in res.items()}, "verdicts", vd, flush=True)
    # escapes per build workgroup, from a table built into a buffer of our own
    E, d = 1 << (pw - 2), dl.value
    al = lambda v: (v + 255) & ~255
    lg = 14 if E >= (1 << 24) else 12
    n_wg = (E >> 1) >> lg
    esc_off = al(E) + al((E >> d) * 16)
    total = esc_off + al(n_wg * 128 * 16) + 256
    ws = torch.zeros(total, dtype=torch.uint8, device="cuda")
    flag = ctypes.c_uint32(7)
    L.bhw_dbg_check_table_format.argtypes = [ctypes.POINTER(B.BhwParams), ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
    rc = L.bhw_dbg_check_table_format(ctypes.byref(p), 0, None, 48 + d, ctypes.c_void_p(ws.data_ptr()), ctypes.byref(flag))
    torch.cuda.synchronize()
    lists = ws[esc_off:esc_off + n_wg * 128 * 16].view(torch.int32).view(n_wg, 128 * 4)
    cnt = (lists.view(n_wg, 128, 4)[:, :, 0] != -1).sum(dim=1).cpu()
    print("   escape tables: rc", rc, "overflow flag", flag.value, "workgroups", n_wg, "total", int(cnt.sum()), "max per workgroup", int(cnt.max()), flush=True)
