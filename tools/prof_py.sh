#!/bin/bash
# rocprofv3 kernel-trace of a python script: bash tools/prof_py.sh <tag> <script.py>
tag=$1; script=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python $script > gpurun_out/prof_$tag.log 2>&1
python - <<PY
import csv,glob
for f in glob.glob("gpurun_out/prof_$tag/*/*kernel_stats.csv"):
    for row in list(csv.reader(open(f)))[1:8]:
        print("%-70s calls %4s avg %8.1f us" % (row[0][:70], row[1], float(row[3])/1e3))
PY
