#!/bin/bash
# Round artefacts in one GPU call: full GPU suite, hardware-counter passes (headline + every extra leg), bench (default + driver-style
# short runs), rocprofv3 kernel stats (headline step alone, and the default command with all legs).
# usage (GPU box, repo root): bash tools/gpu_round.sh <tag>
tag=${1:-r05}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/pytest_$tag.log 2>&1
rc=$?; tail -3 gpurun_out/pytest_$tag.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)" gpurun_out/pytest_$tag.log | head -20; exit $rc; }
# counters first: bench.py cites profiles/pmc_latest.json / pmc_legs_latest.json and withholds them when the kernel sources are newer
bash tools/gpu_pmc.sh $tag > gpurun_out/pmc_$tag.txt 2>&1 && cp gpurun_out/pmc_latest.json profiles/pmc_latest.json
tail -1 gpurun_out/pmc_$tag.txt
bash tools/gpu_pmc_legs.sh $tag > gpurun_out/pmc_legs_$tag.txt 2>&1 && cp gpurun_out/pmc_legs_latest.json profiles/pmc_legs_latest.json
tail -10 gpurun_out/pmc_legs_$tag.txt
timeout -k 10 300 python bench.py > gpurun_out/bench_${tag}_default.json 2> gpurun_out/bench_${tag}_default.err || exit 1
for i in 1 2; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_${tag}_short$i.json 2>> gpurun_out/bench_${tag}_default.err || exit 1
done
rm -rf gpurun_out/prof_$tag gpurun_out/prof_${tag}_all
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs --no-cpp-leg > gpurun_out/bench_${tag}_headline_under_rocprof.json 2> gpurun_out/bench_${tag}_rocprof.err || exit 1
cp gpurun_out/prof_$tag/*/*kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_all -- python bench.py --no-cpu-baseline > gpurun_out/bench_${tag}_under_rocprof.json 2>> gpurun_out/bench_${tag}_rocprof.err || exit 1
cp gpurun_out/prof_${tag}_all/*/*kernel_stats.csv gpurun_out/${tag}_kernel_stats_all_legs.csv
python - <<PY
import json
for n in ("default","short1","short2","headline_under_rocprof","under_rocprof"):
    r=json.load(open("gpurun_out/bench_${tag}_%s.json"%n))
    print("%-22s value %.1f Gs/s  ms/step %.4f  frac %.4f  dev %.4f  no-ramp %.4f parity %s" % (n, r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["device_ms_per_step"], r["ramp"]["no_ramp"]["ms_per_step"], r["parity_spot_check"]))
PY
head -6 gpurun_out/${tag}_kernel_stats.csv
