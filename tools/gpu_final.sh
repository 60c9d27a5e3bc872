#!/bin/bash
# end-of-round measurement set: gpu tests, default bench line, rocprof kernel stats (train step, supervision, full-res CRF),
# PMC passes (HBM traffic, LDS counters, MFMA busy).   usage: bash tools/gpu_final.sh outdir
OUT=${1:-gpurun_out/final}
mkdir -p $OUT
export PYTHONPATH=$PWD
ROOT=$PWD
timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o train -- python $ROOT/bench.py --steps 20 --warmup 8 --no-fp32 --no-cpu-baseline --no-modes --no-profile > $ROOT/$OUT/bench_train_rocprof.json 2> $ROOT/$OUT/rocprof_train.err
cp /tmp/prof_t/train_results.db $ROOT/$OUT/ 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o sup -- python $ROOT/bench.py --mode supervision --steps 50 --warmup 10 --no-cpu-baseline > $ROOT/$OUT/bench_sup_rocprof.json 2> $ROOT/$OUT/rocprof_sup.err
cp /tmp/prof_s/sup_results.db $ROOT/$OUT/ 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o fr -- python $ROOT/bench.py --mode crf-fullres --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/$OUT/bench_fr_rocprof.json 2> $ROOT/$OUT/rocprof_fr.err
cp /tmp/prof_f/fr_results.db $ROOT/$OUT/ 2>/dev/null
cd $ROOT
python tools/rocpd_stats.py $OUT/train_results.db 70 20 avgpool3x3_s1 2 > $OUT/train_kernel_stats.txt 2>&1
python tools/rocpd_stats.py $OUT/sup_results.db 40 20 sup_grad_kernel > $OUT/sup_kernel_stats.txt 2>&1
python tools/rocpd_stats.py $OUT/fr_results.db 40 > $OUT/fullres_kernel_stats.txt 2>&1
bash tools/gpu_pmc.sh $OUT/pmc sup_fetch sup_write sup_lds fr_fetch fr_write train_mfma
for B in 16 1; do python tools/sup_graph_probe.py $B 2>&1 | grep -v amdgpu; done > $OUT/probe.txt
{ timeout 400 python tools/parity_sweep.py 30; timeout 400 python tools/parity_sweep_crf.py 60; timeout 600 python tools/parity_sweep_shapes.py 40; } 2>&1 | grep -v amdgpu > $OUT/parity_sweeps.txt
python tools/filter_trace.py 16 2>&1 | grep -v amdgpu > $OUT/filter_trace.txt
python tools/build_trace.py 16 2>&1 | grep -v amdgpu > $OUT/build_trace.txt
head -3 $OUT/train_kernel_stats.txt | cut -c1-200; cat $OUT/probe.txt; tail -12 $OUT/parity_sweeps.txt
