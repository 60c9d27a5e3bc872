#!/bin/bash
# A/B of library builds (DSRG_LIB) on one box: quick parity check per build, then phase trace + supervision bench.
# usage: bash tools/gpu_ab_lib.sh outdir lib1 lib2 ...   (names relative to dsrg_amd/, "libdsrg_hip.so" = the shipped build)
OUT=$1; shift
mkdir -p $OUT
export PYTHONPATH=$PWD
for lib in "$@"; do
  tag=${lib#libdsrg_hip.}; tag=${tag%.so}; [ "$tag" = "so" ] && tag=base; [ -z "$tag" ] && tag=base
  echo "=== $lib" | tee -a $OUT/trace.log >> $OUT/ab.log
  DSRG_LIB=$lib timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "filter_variants or single_filter or crf_refine_batch or fused_step_matches" > $OUT/pytest_$tag.log 2>&1
  echo "pytest rc=$? $(tail -1 $OUT/pytest_$tag.log)" >> $OUT/ab.log
  DSRG_LIB=$lib timeout 120 python tools/filter_trace.py 16 >> $OUT/trace.log 2>&1
  for rep in 1 2; do
  for B in 16 1; do
    DSRG_LIB=$lib timeout 200 python bench.py --mode supervision --steps 50 --warmup 10 --batch $B --no-cpu-baseline > $OUT/sup_${tag}_b${B}.json 2>$OUT/sup_${tag}_b${B}.err
    python - <<PY >> $OUT/ab.log
import json
try:
    j=json.load(open("$OUT/sup_${tag}_b${B}.json"))
    r=j["roofline"]
    print("%-8s B %2d  ms/step %.4f  filter us/launch %.2f  frac %.3f" % ("$tag", $B, j["ms_per_step"], r["us_per_launch"], r["frac"]))
except Exception as e:
    print("$tag B $B failed", e)
PY
  done
  done
done
cat $OUT/ab.log
