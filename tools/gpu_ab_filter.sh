#!/bin/bash
# A/B of the mean-field filter options on one box: parity tests first, then phase traces and the supervision bench per option set.
# usage (on the GPU box, from the repo root): bash tools/gpu_ab_filter.sh [outdir]
OUT=${1:-gpurun_out/ab_filter}
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "srg or filter or lattice or crf or fused or softmax or losses or error_paths" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
echo "=== build trace" >> $OUT/trace.log; timeout 120 python tools/build_trace.py 16 >> $OUT/trace.log 2>&1
for o in 0 1 3 5 9 15; do
  echo "=== DSRG_FILTER_OPTS=$o" >> $OUT/trace.log
  DSRG_FILTER_OPTS=$o timeout 120 python tools/filter_trace.py 16 >> $OUT/trace.log 2>&1
  for B in 16 1; do
    DSRG_FILTER_OPTS=$o timeout 200 python bench.py --mode supervision --steps 50 --warmup 10 --batch $B --no-cpu-baseline > $OUT/sup_o${o}_b${B}.json 2>$OUT/sup_o${o}_b${B}.err
    python - <<PY >> $OUT/ab.log
import json
try:
    j=json.load(open("$OUT/sup_o${o}_b${B}.json"))
    r=j["roofline"]
    print("opts %2d B %2d  ms/step %.4f  filter us/launch %.2f  frac %.3f" % ($o, $B, j["ms_per_step"], r["us_per_launch"], r["frac"]))
except Exception as e:
    print("opts $o B $B failed", e)
PY
  done
done
cat $OUT/ab.log
