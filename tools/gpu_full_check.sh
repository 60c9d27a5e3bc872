#!/bin/bash
# full GPU check of the tree: gpu test suite, default bench line, rocprof kernel stats of the supervision path
# usage: bash tools/gpu_full_check.sh outdir
OUT=${1:-gpurun_out/full}
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o sup -- python $ROOT/bench.py --mode supervision --steps 50 --warmup 10 --no-cpu-baseline > $ROOT/$OUT/bench_sup_rocprof.json 2> $ROOT/$OUT/rocprof.err
cp /tmp/prof/sup_results.db $ROOT/$OUT/ 2>/dev/null
cd $ROOT
python tools/rocpd_stats.py $OUT/sup_results.db 40 20 sup_grad_kernel > $OUT/sup_kernel_stats.txt 2>&1
head -30 $OUT/sup_kernel_stats.txt
python - <<PY
import json
j=json.load(open("$OUT/bench_default.json"))
print({k: j[k] for k in ("value","ms_per_step","value_fp32","supervision_ms_per_step")})
print("roofline", {k: j["roofline"][k] for k in ("us_per_launch","frac","achieved","peak")})
for m,r in j.get("modes",{}).items():
    print(m, {k: r.get(k) for k in ("value","ms_per_step","error")}, (r.get("roofline") or {}).get("frac"), (r.get("cpu_baseline") or {}).get("value"))
PY
