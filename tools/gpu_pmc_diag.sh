#!/bin/bash
# Wide hardware-counter diagnosis of the headline step (build + combine kernels): issue, instruction fetch, vector-memory path,
# address translation, L2 and fabric stalls.  One rocprofv3 --pmc pass per set (kernel-trace only).
# (TA_* and TCP_* counters, and more than four TCC counters in one pass, abort rocprofv3 on this pool and then hang until the timeout.  BHW_DIAG_SETS=4,5: only those passes)
# usage (GPU box, repo root): bash tools/gpu_pmc_diag.sh <tag> [extra bench.py args]
tag=${1:-diag}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  [ -n "$BHW_DIAG_SETS" ] && ! echo ",$BHW_DIAG_SETS," | grep -q ",$i," && continue
  rm -rf gpurun_out/pmcd_${tag}_$i
  timeout -k 5 40 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcd_${tag}_$i -- python bench.py --steps 3 --warmup 1 --ramp-seconds 0.2 --no-cpu-baseline --no-extra-legs --no-cpp-leg "$@" > gpurun_out/pmcd_${tag}_$i.json 2> gpurun_out/pmcd_${tag}_$i.err || echo "pass $i failed: $set"
done <<SETS
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_LEVEL_WAVES SQ_CYCLES SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM GRBM_GUI_ACTIVE
SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAVES
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
TCC_REQ_sum TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum
TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum
TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum
TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum
SETS
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcd_${tag}_*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
        agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in agg.items():
    if not k.startswith("k_"): continue
    print(k)
    for c,vals in sorted(v.items()): print("   %-40s n=%3d  mean=%.6g" % (c,len(vals),sum(vals)/len(vals)))
PY
