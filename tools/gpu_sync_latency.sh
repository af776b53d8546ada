#!/bin/bash
# Driver-style short runs (--steps 20 --warmup 5) under different host wait modes; and the short-launch floor.
set -e
mkdir -p gpurun_out
O=gpurun_out/sync_latency.txt
: > $O
run() { # label, env...
  local label=$1; shift
  for i in 1 2 3; do
    env "$@" python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-cpp-leg 2>/dev/null | tail -1 \
      | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$label', r['ms_per_step'], r.get('device_ms_per_step'))" >> $O
  done
}
run default A=1
run hsa_poll HSA_ENABLE_INTERRUPT=0
run roc_active_wait ROC_ACTIVE_WAIT_TIMEOUT=200
run both HSA_ENABLE_INTERRUPT=0 ROC_ACTIVE_WAIT_TIMEOUT=200
cat $O
timeout -k 10 120 build/ubench_short | tee gpurun_out/ubench_short.txt
