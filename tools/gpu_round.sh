#!/bin/bash
# Round artefacts in one GPU call: full GPU suite, bench (default + driver-style short runs), rocprofv3 kernel stats, PMC passes.
# usage (GPU box, repo root): bash tools/gpu_round.sh <tag>
tag=${1:-r02}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/pytest_$tag.log 2>&1
rc=$?; tail -3 gpurun_out/pytest_$tag.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)" gpurun_out/pytest_$tag.log | head -20; exit $rc; }
# counters first: bench.py cites profiles/pmc_latest.json for roofline.traffic and withholds it when the kernel sources are newer
bash tools/gpu_pmc.sh $tag > gpurun_out/pmc_$tag.txt 2>&1 && cp gpurun_out/pmc_latest.json profiles/pmc_latest.json
timeout -k 10 300 python bench.py > gpurun_out/bench_${tag}_default.json 2> gpurun_out/bench_${tag}_default.err || exit 1
for i in 1 2; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_${tag}_short$i.json 2>> gpurun_out/bench_${tag}_default.err || exit 1
done
rm -rf gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_${tag}_under_rocprof.json 2> gpurun_out/bench_${tag}_rocprof.err || exit 1
cp gpurun_out/prof_$tag/*/*kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
python - <<PY
import json
for n in ("default","short1","short2","under_rocprof"):
    r=json.load(open("gpurun_out/bench_${tag}_%s.json"%n))
    print("%-14s value %.1f Gs/s  ms/step %.4f  frac %.4f  dev %.4f  spread %s parity %s" % (n, r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["device_ms_per_step"], {k:round(v,4) for k,v in r["roofline"]["device_ms_per_step_spread"].items()}, r["parity_spot_check"]))
PY
head -8 gpurun_out/${tag}_kernel_stats.csv
tail -3 gpurun_out/pmc_$tag.txt
