#!/bin/bash
# quick check of the supervision path: hot-path parity tests, eager ms/step (B=16, 1), rocprof kernel stats per step
# usage: bash tools/gpu_sup_stats.sh outdir [batch]
OUT=${1:-gpurun_out/sup}
BATCH=${2:-16}
mkdir -p $OUT
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "srg or filter or lattice or crf or fused or golden or glue" > $OUT/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
for B in 16 1; do python tools/sup_graph_probe.py $B 2>&1 | grep -v amdgpu; done | tee $OUT/probe.txt
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o sup -- python $ROOT/bench.py --mode supervision --steps 50 --warmup 10 --batch $BATCH --no-cpu-baseline > $ROOT/$OUT/bench_sup_rocprof.json 2> $ROOT/$OUT/rocprof.err
cp /tmp/prof/sup_results.db $ROOT/$OUT/ 2>/dev/null
cd $ROOT
python tools/rocpd_stats.py $OUT/sup_results.db 40 20 sup_grad_kernel > $OUT/sup_kernel_stats.txt 2>&1
head -18 $OUT/sup_kernel_stats.txt | cut -c1-150
