#!/bin/bash
# Hardware-counter passes for the headline bench (each --pmc set in its own run; kernel-trace only).
# usage: bash tools/gpu_pmc.sh <tag>
tag=${1:-pmc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_${tag}_$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_${tag}_$i -- python bench.py --steps 3 --warmup 1 --ramp-seconds 0.2 --no-cpu-baseline --no-extra-legs > gpurun_out/pmc_${tag}_$i.json 2> gpurun_out/pmc_${tag}_$i.err || echo "pass $i failed"
done
python - <<PY
import csv,glob,collections,json
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_${tag}_*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]      # kernel<template args>
        agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
import json
summary={}
for k,v in agg.items():
    print(k)
    summary[k]={}
    for c,vals in sorted(v.items()):
        print("   %-24s n=%3d  mean=%.4g" % (c,len(vals),sum(vals)/len(vals)))
        summary[k][c]={"launches":len(vals),"mean":sum(vals)/len(vals)}
# HBM traffic per launch, MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts
# 128-B requests as 64 B -> doubled (cross-checked against TCC_EA0_RDREQ_sum x 128 B); WRITE_SIZE is exact (calibrated
# on the table-build kernel, which writes exactly 8 B x entries).
step=0.0
# the headline step = the kernels bench.py's plan line names (the same run also times the model-cpp leg: its kernels are listed
# with their own counters but do not belong to the step)
plan=json.load(open("gpurun_out/pmc_${tag}_1.json"))["config"]["plan"]
heads=[s.strip().split(" ")[0].replace(",", ", ").replace(">", "") for s in plan.split(": ",1)[1].split(" + ")]
for k,v in summary.items():
    if "bhw" not in k.lower() and "k_table" not in k and "k_direct" not in k and "k_fold" not in k and "k_tile9" not in k: continue   # only this library's kernels
    f=2*1024*v.get("FETCH_SIZE",{}).get("mean",0.0); w=1024*v.get("WRITE_SIZE",{}).get("mean",0.0)
    v["hbm_bytes_per_launch"]={"read":f,"write":w,"total":f+w}
    if any(k.startswith(h) for h in heads):
        v["in_headline_step"]=True
        step+=f+w
summary["_step_hbm_bytes"]=step
import datetime, sys, os
sys.path.insert(0, os.getcwd())
import bench
summary["_meta"]={"date": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"), "sources_sha16": bench.sources_sha16(),
                  "command": "rocprofv3 --kernel-trace --pmc <set> -- python bench.py --steps 3 --warmup 1 --ramp-seconds 0.2 --no-cpu-baseline --no-extra-legs (6 passes; the model-cpp leg runs too)", "headline_kernels": heads}
json.dump(summary,open("gpurun_out/pmc_summary_${tag}.json","w"),indent=1,sort_keys=True)
json.dump(summary,open("gpurun_out/pmc_latest.json","w"),indent=1,sort_keys=True)
print("step HBM bytes: %.1f MB" % (step/1e6))
PY
