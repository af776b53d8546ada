#!/bin/bash
# Hardware-counter passes for the headline bench (each --pmc set in its own run; kernel-trace only).
# usage: bash tools/gpu_pmc.sh <tag>
tag=${1:-pmc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_${tag}_$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_${tag}_$i -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_${tag}_$i.json 2> gpurun_out/pmc_${tag}_$i.err || echo "pass $i failed"
done
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_${tag}_*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()):
        print("   %-24s n=%3d  mean=%.4g" % (c,len(vals),sum(vals)/len(vals)))
PY
