#!/bin/bash
# TA/TCP counter passes for the headline bench.  usage: bash tools/gpu_pmc2.sh <tag> [extra build flags]
tag=${1:-ta}; flags="$2"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BHW_EXTRA_FLAGS="$flags" python -c "from blackman_harris_win_amd import _build; _build.build_library(force=True)" > /dev/null 2>&1
i=0
for set in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1)); rm -rf gpurun_out/pmc2_${tag}_$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc2_${tag}_$i -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc2_${tag}_$i.err || echo "pass $i failed"
done
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc2_${tag}_*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:48]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in agg.items():
    if "k_table" not in k: continue
    print(k)
    for c,vals in sorted(v.items()):
        print("   %-40s %.4g" % (c,sum(vals)/len(vals)))
PY
