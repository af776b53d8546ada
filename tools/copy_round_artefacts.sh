#!/bin/bash
# After tools/gpu_round.sh <tag> has been merged back into gpurun_out/: copy what DESIGN.md / README.md cite into profiles/ (tracked).
# usage (CPU container, repo root): bash tools/copy_round_artefacts.sh <tag>
tag=${1:-r05}
set -e
cd "$(dirname "$0")/.."
g=gpurun_out
cp $g/bench_${tag}_default.json profiles/${tag}_bench_default.json
cp $g/bench_${tag}_short1.json profiles/${tag}_bench_steps20_a.json
cp $g/bench_${tag}_short2.json profiles/${tag}_bench_steps20_b.json
cp $g/bench_${tag}_headline_under_rocprof.json profiles/${tag}_bench_headline_under_rocprof.json
cp $g/bench_${tag}_under_rocprof.json profiles/${tag}_bench_under_rocprof.json
cp $g/${tag}_kernel_stats.csv $g/${tag}_kernel_stats_all_legs.csv profiles/
cp $g/pmc_summary_${tag}.json profiles/${tag}_pmc_summary.json
cp $g/pmc_legs_latest.json profiles/${tag}_pmc_legs.json
cp $g/pmc_latest.json $g/pmc_legs_latest.json profiles/
cp blackman_harris_win_amd/kernel_resources.json profiles/${tag}_kernel_resources.json
python - <<PY
import bench, json
print("sources", bench.sources_sha16(), "pmc", json.load(open("profiles/pmc_latest.json"))["_meta"]["sources_sha16"],
      "legs", json.load(open("profiles/pmc_legs_latest.json"))["_meta"]["sources_sha16"])
PY
