#!/bin/bash
# Round-end artefacts on the GPU box: GPU suite, default bench, the same command under rocprofv3 --stats, PMC passes, other configs.
# usage: bash tools/gpu_final.sh <tag>
tag=${1:-final}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$tag.log 2>&1 || { tail -20 gpurun_out/pytest_$tag.log; exit 1; }
tail -1 gpurun_out/pytest_$tag.log
timeout -k 10 300 python bench.py > gpurun_out/bench_${tag}_default.json 2> gpurun_out/bench_${tag}_default.err || exit 2
echo "default bench done"
rm -rf gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --no-cpu-baseline > gpurun_out/bench_${tag}_under_rocprof.json 2> gpurun_out/bench_${tag}_under_rocprof.err || exit 3
echo "rocprof bench done"
cp gpurun_out/prof_$tag/*/*kernel_stats.csv gpurun_out/kernel_stats_$tag.csv
bash tools/gpu_pmc.sh $tag > gpurun_out/pmc_$tag.txt 2>&1
tail -1 gpurun_out/pmc_$tag.txt
timeout -k 10 400 python tools/bench_configs.py > gpurun_out/other_configs_$tag.json 2> gpurun_out/other_configs_$tag.err || exit 4
echo "other configs done"
python - <<PY
import json
for f in ("default", "under_rocprof"):
    r = json.load(open("gpurun_out/bench_${tag}_%s.json" % f))
    print(f, "%.1f Gsamples/s  %.4f ms  frac %.4f" % (r["value"], r["ms_per_step"], r["roofline"]["frac"]), r["roofline"]["per_kernel"])
PY
head -4 gpurun_out/kernel_stats_$tag.csv
