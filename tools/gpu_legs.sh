#!/bin/bash
# GPU parity suite, then the default bench under rocprofv3 --stats (all legs): kernel stats + the legs' numbers.
# usage (GPU box, repo root): bash tools/gpu_legs.sh <tag>
tag=${1:-legs}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$tag.log 2>&1
  rc=$?; tail -3 gpurun_out/pytest_$tag.log
  [ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR|E )" gpurun_out/pytest_$tag.log | head -20; exit $rc; }
fi
rm -rf gpurun_out/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --no-cpu-baseline > gpurun_out/bench_${tag}_rocprof.json 2> gpurun_out/bench_${tag}_rocprof.err || { tail -5 gpurun_out/bench_${tag}_rocprof.err; exit 1; }
cp gpurun_out/prof_$tag/*/*kernel_stats.csv gpurun_out/${tag}_kernel_stats_all_legs.csv
python - <<PY
import json,csv
r=json.load(open("gpurun_out/bench_${tag}_rocprof.json"))
print("headline %.1f Gs/s  %.4f ms  frac %.4f  no-ramp %.4f" % (r["value"], r["ms_per_step"], r["roofline"]["frac"], r["ramp"]["no_ramp"]["ms_per_step"]))
print("cpp", r["cpp_model"]["ms_per_step_median"])
for k,v in r["extra_legs"].items():
    if isinstance(v, dict) and "ms" in v: print("  %-36s %.4f ms  %s" % (k, v["ms"], v.get("plan","")[:90]))
for row in list(csv.reader(open("gpurun_out/${tag}_kernel_stats_all_legs.csv")))[1:14]:
    print("   %-70s calls %6s avg %8.1f us" % (row[0].replace("void (anonymous namespace)::","")[:70], row[1], float(row[3])/1e3))
PY
