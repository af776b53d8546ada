#!/usr/bin/env python3
"""L2 request rates of the long-window kernels, from the committed per-leg counter passes (profiles/pmc_legs_latest.json):
requests = TCC_HIT_sum + TCC_MISS_sum per launch; cycles = GRBM_GUI_ACTIVE per launch, which rocprofv3 sums over the eight XCDs
(cycles / duration = 18 .. 24 "GHz" in the same passes), so cycles / 8 is the kernel's length in shader-engine clocks.
usage: python tools/l2_request_rates.py [pmc_legs.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "pmc_legs_latest.json")
d = json.load(open(path))
print("%-26s %-46s %9s %9s %9s %10s %9s" % ("leg", "kernel", "requests", "written", "read", "clocks", "req/clock"))
for leg in sorted(k for k in d if not k.startswith("_")):
    for name, c in d[leg]["kernels"].items():
        req = c.get("l2_hit", 0.0) + c.get("l2_miss", 0.0)
        clk = c.get("gui_active_cycles", 0.0) / 8.0
        if req < 1e6:
            continue                                       # short kernels: launch-bound, not request-bound
        print("%-26s %-46s %8.2fM %6.0f MB %6.0f MB %10.0f %9.1f" % (leg, name[:46], req / 1e6, c.get("write_bytes", 0) / 1e6, c.get("read_bytes", 0) / 1e6, clk, req / clk))
