#!/usr/bin/env python3
"""Groups a rocprofv3 kernel_trace.csv by (kernel, grid size): calls, mean and min duration in microseconds."""
import collections
import csv
import glob
import re
import sys

rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"^void \(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))
        rows[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), v in sorted(rows.items()):
    if name.startswith("k_"):
        v = sorted(v)
        print("%-44s grid %10d  calls %4d  mean %9.2f us  median %9.2f  min %9.2f" % (name[:44], grid, len(v), sum(v) / len(v), v[len(v) // 2], v[0]))
