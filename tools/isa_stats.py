#!/usr/bin/env python3
"""Resource usage and instruction mix of the kernels in an ISA listing of one kernel translation unit (csrc/bhw_*.hip).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S blackman_harris_win_amd/csrc/bhw_build.hip -o /tmp/k.s
    python tools/isa_stats.py /tmp/k.s 'k_table_combine_tile<15, 0, 2>' [--mix]
"""
import collections
import re
import subprocess
import sys


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    return out[:len(names)]


def main():
    path, pats = sys.argv[1], [a for a in sys.argv[2:] if not a.startswith("--")]
    mix = "--mix" in sys.argv
    t = open(path).read()
    blocks = re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", t, re.S)
    names = [b[0] for b in blocks]
    dem = demangle(names)
    for (n, body), d in zip(blocks, dem):
        short = re.sub(r"^void \(anonymous namespace\)::", "", d).split("(")[0]
        if pats and not any(p in short for p in pats):
            continue
        g = lambda f: re.search(r"\.amdhsa_%s (\S+)" % f, body).group(1)  # noqa: E731
        # code of this function: from its label to the end of the "; Kernel info" comment block that follows it
        i0 = t.find("\n%s:" % n)
        i1 = t.find("; Occupancy:", i0)
        i1 = t.find("\n", i1) if i1 >= 0 else -1
        text = t[i0:i1] if i0 >= 0 and i1 >= 0 else ""
        spill = re.search(r"; ScratchSize: (\d+)", text)
        occ = re.search(r"; Occupancy: (\d+)", text)
        vg = re.search(r"; NumVgprs: (\d+)", text)
        sg = re.search(r"; TotalNumSgprs: (\d+)", text)
        code = re.search(r"; codeLenInByte = (\d+)", text)
        print("%-52s vgpr %s sgpr %s lds %s scratch %s occupancy %s code %s B" % (
            short[:52], vg and vg.group(1), sg and sg.group(1), g("group_segment_fixed_size"),
            spill and spill.group(1), occ and occ.group(1), code and code.group(1)))
        if mix and text:
            ops = collections.Counter()
            for line in text.split("\n"):
                mm = re.match(r"\s+([a-z_0-9]+)\s", line)
                if mm and not mm.group(1).startswith("."):
                    ops[mm.group(1)] += 1
            tot = sum(ops.values())
            valu = sum(v for k, v in ops.items() if k.startswith("v_"))
            print("   static instructions %d, VALU %d, SALU %d, vmem %d, lds %d" % (
                tot, valu, sum(v for k, v in ops.items() if k.startswith("s_")),
                sum(v for k, v in ops.items() if k.startswith(("global_", "buffer_", "flat_"))),
                sum(v for k, v in ops.items() if k.startswith("ds_"))))
            print("   " + ", ".join("%s %d" % kv for kv in ops.most_common(28)))


if __name__ == "__main__":
    main()
