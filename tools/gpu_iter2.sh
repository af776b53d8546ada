#!/bin/bash
# GPU suite, then the default bench (extra legs included); prints the legs.   usage: bash tools/gpu_iter2.sh <tag>
tag=${1:-it}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/pytest_$tag.log 2>&1
rc=$?; tail -3 gpurun_out/pytest_$tag.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR|E )" gpurun_out/pytest_$tag.log | head -20; exit $rc; }
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { tail -5 gpurun_out/bench_$tag.err; exit 2; }
python - <<PY
import json
r = json.loads(open("gpurun_out/bench_$tag.json").read().strip().split("\n")[-1])
print("headline %.1f Gsamples/s  %.4f ms  frac %.4f" % (r["value"], r["ms_per_step"], r["roofline"]["frac"]))
for k, v in r["extra_legs"].items():
    if isinstance(v, dict): print("  %-28s %.4f ms  %s" % (k, v.get("ms", 0), v.get("plan", "")))
PY
