#!/bin/bash
# One GPU iteration: GPU parity suite, in-process A/B of the current sources against prebuilt libraries, rocprofv3 kernel stats.
# usage (GPU box, repo root): bash tools/gpu_ab.sh <tag> "<flags A>" ["<flags B>" ...]      env: AB_EXTRA_LIBS, SKIP_TESTS=1
tag=${1:-ab}; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/pytest_$tag.log 2>&1
  rc=$?; tail -3 gpurun_out/pytest_$tag.log
  [ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR|E )" gpurun_out/pytest_$tag.log | head -20; exit $rc; }
fi
timeout -k 10 300 python tools/ab_inproc.py "$@" > gpurun_out/ab_$tag.txt 2>&1 || { tail -20 gpurun_out/ab_$tag.txt; exit 1; }
cat gpurun_out/ab_$tag.txt
rm -rf gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { tail -5 gpurun_out/bench_$tag.err; exit 1; }
python - <<PY
import json,glob,csv
r=json.load(open("gpurun_out/bench_$tag.json"))
print("value %.2f Gsamples/s  ms/step %.4f  roofline frac %.4f parity %s" % (r["value"], r["ms_per_step"], r["roofline"]["frac"], r["parity_spot_check"]))
for f in glob.glob("gpurun_out/prof_$tag/*/*kernel_stats.csv"):
    for row in list(csv.reader(open(f)))[1:7]:
        print("   %-60s calls %4s avg %10.1f us  %5s%%" % (row[0][:60], row[1], float(row[3])/1e3, row[4]))
PY
