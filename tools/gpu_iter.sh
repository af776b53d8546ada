#!/bin/bash
# One GPU iteration: full GPU parity suite, then a rocprofv3 kernel-trace of the headline bench.
# usage (on the GPU box, from the repo root): bash tools/gpu_iter.sh <tag> [bench args]
tag=${1:-iter}; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/pytest_$tag.log 2>&1
rc=$?; tail -3 gpurun_out/pytest_$tag.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR|E )" gpurun_out/pytest_$tag.log | head -20; exit $rc; }
rm -rf gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
echo "bench rc=$?"
python - <<PY
import json,glob
r=json.load(open("gpurun_out/bench_$tag.json"))
print("value %.2f Gsamples/s  ms/step %.4f  roofline frac %.4f parity %s" % (r["value"], r["ms_per_step"], r["roofline"]["frac"], r["parity_spot_check"]))
import csv
for f in glob.glob("gpurun_out/prof_$tag/*/*kernel_stats.csv"):
    for row in list(csv.reader(open(f)))[1:7]:
        print("   %-60s calls %4s avg %10.1f us  %5s%%" % (row[0][:60], row[1], float(row[3])/1e3, row[4]))
PY
