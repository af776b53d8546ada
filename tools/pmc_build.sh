cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in 0 1; do
  rm -rf gpurun_out/pmc_bm$m
  BHW_BUILD_MIRROR=$m timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_bm$m -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_bm$m/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "k_table_build" in row["Kernel_Name"]: agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("mirror=$m", {k: "%.3g" % (sum(v)/len(v)) for k,v in sorted(agg.items())})
PY
done
