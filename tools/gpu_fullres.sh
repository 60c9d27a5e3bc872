#!/bin/bash
# full-resolution CRF path: parity tests + bench + rocprof kernel stats
OUT=${1:-gpurun_out/fullres}
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests -x -q -m gpu -k "large or fullres or 321 or 375 or krahenbuhl or crf_function or inference or predict" > $OUT/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
timeout 300 python bench.py --mode crf-fullres --steps 20 --warmup 5 > $OUT/bench_fullres.json 2> $OUT/bench.err
python - <<PY
import json
j=json.load(open("$OUT/bench_fullres.json"))
print({k: j[k] for k in ("value","ms_per_step","max_abs_dq_vs_oracle")}, {k: j["roofline"][k] for k in ("us_per_launch","frac","achieved")})
for s in j["sizes"]: print(s["H"], s["W"], s["ms_per_image"], s["M_gauss"], s["M_bil"])
PY
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o fr -- python $ROOT/bench.py --mode crf-fullres --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $ROOT/$OUT/rocprof.err
cp /tmp/prof/fr_results.db $ROOT/$OUT/ 2>/dev/null
cd $ROOT
python tools/rocpd_stats.py $OUT/fr_results.db 30 > $OUT/fullres_kernel_stats.txt 2>&1
head -24 $OUT/fullres_kernel_stats.txt | cut -c1-160
