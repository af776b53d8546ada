#!/bin/bash
# Hardware-counter passes for every extra leg of bench.py (tools/pmc_legs.py): bytes written / fetched and instruction counts per
# launch of each leg's kernels -> gpurun_out/pmc_legs_latest.json (copy to profiles/ and commit; bench.py cites it per leg).
# Each --pmc set in its own run, kernel-trace only (the pool refuses --pmc with the API trace domains).
# usage: bash tools/gpu_pmc_legs.sh <tag>
tag=${1:-legs}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  j=0
  for leg in $(python -c "import sys; sys.path.insert(0, 'tools'); import pmc_legs; print(' '.join(pmc_legs.LEGS))"); do
    j=$((j+1))
    rm -rf gpurun_out/pmcl_${tag}_${i}_$j
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcl_${tag}_${i}_$j -- python tools/pmc_legs.py "$leg" > gpurun_out/pmcl_${tag}_${i}_$j.json 2> gpurun_out/pmcl_${tag}_${i}_$j.err || echo "pass $i leg $leg failed"
  done
  echo "counter set $i done"
done
python - <<PY
import csv, glob, collections, json, datetime, sys, os
sys.path.insert(0, os.getcwd())
import bench
out = {}
sys.path.insert(0, "tools")
import pmc_legs as PL
for j in range(1, len(PL.LEGS) + 1):
    legs = json.loads([l for l in open("gpurun_out/pmcl_${tag}_1_%d.json" % j) if l.startswith("{")][-1])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/pmcl_${tag}_*_%d/*/*counter_collection.csv" % j):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if name.startswith("void "): name = name[5:]
            name = name.replace("(anonymous namespace)::", "", 1).split("(")[0]
            agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for leg, info in legs.items():
        ks = {}
        tot_w = tot_r = 0.0
        for pref in info["kernels"]:
            for name, ctr in agg.items():
                if not name.startswith(pref):
                    continue
                m = {c: sum(v) / len(v) for c, v in ctr.items()}
                # MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B -> doubled
                w = 1024.0 * m.get("WRITE_SIZE", 0.0)
                r = 2.0 * 1024.0 * m.get("FETCH_SIZE", 0.0)
                ks[name] = {"launches_counted": len(ctr.get("WRITE_SIZE", [])), "write_bytes": w, "read_bytes": r,
                            "valu_wave_insts": m.get("SQ_INSTS_VALU"), "salu_wave_insts": m.get("SQ_INSTS_SALU"), "waves": m.get("SQ_WAVES"),
                            "vmem_rd_wave_insts": m.get("SQ_INSTS_VMEM_RD"), "vmem_wr_wave_insts": m.get("SQ_INSTS_VMEM_WR"), "lds_wave_insts": m.get("SQ_INSTS_LDS"),
                            "l2_hit": m.get("TCC_HIT_sum"), "l2_miss": m.get("TCC_MISS_sum"), "gui_active_cycles": m.get("GRBM_GUI_ACTIVE")}
                tot_w += w
                tot_r += r
        out[leg] = {"plan": info["plan"], "algorithmic_bytes": info["algorithmic_bytes"], "kernels": ks,
                    "traffic_bytes_per_call": tot_w + tot_r, "write_bytes_per_call": tot_w, "read_bytes_per_call": tot_r,
                    "traffic_over_algorithmic": (tot_w + tot_r) / info["algorithmic_bytes"] if info["algorithmic_bytes"] else None}
        print("%-28s write %8.1f MB  read %8.1f MB  (algorithmic %8.1f MB)  %s" % (leg, tot_w / 1e6, tot_r / 1e6, info["algorithmic_bytes"] / 1e6, list(ks)))
out["_meta"] = {"date": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"), "sources_sha16": bench.sources_sha16(),
                "command": "rocprofv3 --kernel-trace --pmc <set> -- python tools/pmc_legs.py <leg> (4 counter sets x 10 legs, one process each: several legs launch kernels of the same name)",
                "units": "bytes per launch: WRITE_SIZE KiB x 1024; FETCH_SIZE KiB x 1024 x 2 (gfx950 counts a 128-B request as 64 B)"}
json.dump(out, open("gpurun_out/pmc_legs_latest.json", "w"), indent=1, sort_keys=True)
PY
