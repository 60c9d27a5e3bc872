#!/bin/bash
# A/B of the train step under values of one environment variable, alternating on one box.
# usage: bash tools/gpu_ab_env.sh VAR value1 value2 [reps]
VAR=$1; A=$2; B=$3; REPS=${4:-2}
export PYTHONPATH=$PWD
mkdir -p gpurun_out/ab_env
for rep in $(seq $REPS); do
  for v in $A $B; do
    env $VAR=$v timeout 300 python bench.py --steps 20 --warmup 8 --no-fp32 --no-cpu-baseline --no-modes --no-profile > gpurun_out/ab_env/$v.json 2> gpurun_out/ab_env/$v.err
    python - <<PY
import json
try:
    j=json.load(open("gpurun_out/ab_env/$v.json")); print("$VAR=$v  %.1f img/s  %.3f ms/step  losses %s" % (j["value"], j["ms_per_step"], j["losses"]))
except Exception as e: print("$VAR=$v failed", e)
PY
  done
done
