#!/bin/bash
# rocprofv3 kernel stats of the headline step alone (no extra legs: the fused-apply leg launches the same kernel instance as the
# headline and would be averaged into it), and two driver-style short runs.   usage: bash tools/gpu_stats_headline.sh <tag>
tag=${1:-hs}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --no-cpu-baseline --no-extra-legs --no-cpp-leg > gpurun_out/bench_${tag}_headline_under_rocprof.json 2> gpurun_out/bench_${tag}_headline_under_rocprof.err || exit 3
cp gpurun_out/prof_$tag/*/*kernel_stats.csv gpurun_out/kernel_stats_headline_$tag.csv
head -3 gpurun_out/kernel_stats_headline_$tag.csv | cut -c1-160
for i in a b; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${tag}_steps20_$i.json 2>/dev/null || exit 4
  python -c "import json; r=json.loads(open('gpurun_out/bench_${tag}_steps20_$i.json').read().strip().split('\n')[-1]); print('steps20 $i', r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['traffic'])"
done
