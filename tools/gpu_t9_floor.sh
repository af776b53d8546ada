#!/bin/bash
# k_tile9: timing-only floors (no stores / gathers confined to 64 KiB / no loads), the software-pipelined form, and one PMC pass of the SQ counters.
# usage (GPU box, repo root): bash tools/gpu_t9_floor.sh <tag>      (the variants are prebuilt in the CPU container: tools/ab_inproc.py --build-only)
tag=${1:-t9floor}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AB_UNITS=bhw_tile9.hip AB_NOCHECK=1 AB_SPLIT=1 AB_EXTRA_LIBS=build/ab/libbhw_r4final.so AB_ROUNDS=5
timeout -k 10 500 python tools/ab_inproc.py "" "-DBHW_X_HOT" "-DBHW_X_NOSTORE" "-DBHW_X_HOT -DBHW_X_NOSTORE" "-DBHW_X_NOLOAD" "-DBHW_X_NOLOAD -DBHW_X_NOSTORE" "-DBHW_X_PIPE -DBHW_T9_WAVES=7" "-DBHW_T9_WAVES=7" > gpurun_out/ab_$tag.txt 2>&1 || { tail -20 gpurun_out/ab_$tag.txt; exit 1; }
cat gpurun_out/ab_$tag.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_${tag}_$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_${tag}_$i -- python bench.py --steps 3 --warmup 1 --ramp-seconds 0.2 --no-cpu-baseline --no-extra-legs --no-cpp-leg > gpurun_out/pmc_${tag}_$i.json 2> gpurun_out/pmc_${tag}_$i.err || echo "pass $i failed"
done
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_${tag}_*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
        agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in agg.items():
    if "k_t" not in k: continue
    print(k)
    for c,vals in sorted(v.items()): print("   %-24s n=%3d  mean=%.5g" % (c,len(vals),sum(vals)/len(vals)))
PY
