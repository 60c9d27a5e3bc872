#!/bin/bash
# end-of-round measurement set: gpu tests, default bench line, rocprof kernel stats (train step, supervision, full-res CRF),
# PMC passes (HBM traffic, LDS counters, MFMA busy) parsed into the JSON files bench.py replays, implicit-GEMM probe, parity
# sweeps, phase traces.   usage: bash tools/gpu_final.sh outdir [commit]      (commit: stamped into the counter files)
OUT=${1:-gpurun_out/final}
COMMIT=${2:-unknown}
mkdir -p $OUT
export PYTHONPATH=$PWD
ROOT=$PWD
timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
# counters first: the default bench line below replays them (bench.py drops figures whose kernel sources changed since)
bash tools/gpu_pmc.sh $OUT/pmc sup_fetch sup_write sup_lds fr_fetch fr_write train_mfma
python tools/pmc_parse.py $OUT/pmc/sup_fetch_counter_collection.csv $OUT/pmc/sup_write_counter_collection.csv $OUT/pmc_traffic.json $COMMIT > /dev/null 2>$OUT/pmc_parse.err
python tools/pmc_parse.py $OUT/pmc/fr_fetch_counter_collection.csv $OUT/pmc/fr_write_counter_collection.csv $OUT/pmc_traffic_fullres.json $COMMIT > /dev/null 2>>$OUT/pmc_parse.err
python tools/pmc_lds_parse.py $OUT/pmc/sup_lds_counter_collection.csv $OUT/lds_counters.json $COMMIT > /dev/null 2>>$OUT/pmc_parse.err
python tools/pmc_mfma.py $OUT/pmc/train_mfma_counter_collection.csv 3 > $OUT/mfma_utilisation.txt 2>>$OUT/pmc_parse.err
mkdir -p profiles
for f in pmc_traffic pmc_traffic_fullres lds_counters; do [ -s $OUT/$f.json ] && cp $OUT/$f.json profiles/r06_$f.json; done
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o train -- python $ROOT/bench.py --steps 20 --warmup 8 --no-fp32 --no-cpu-baseline --no-modes --no-profile > $ROOT/$OUT/bench_train_rocprof.json 2> $ROOT/$OUT/rocprof_train.err
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o sup -- python $ROOT/bench.py --mode supervision --steps 50 --warmup 10 --no-cpu-baseline > $ROOT/$OUT/bench_sup_rocprof.json 2> $ROOT/$OUT/rocprof_sup.err
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o fr -- python $ROOT/bench.py --mode crf-fullres --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/$OUT/bench_fr_rocprof.json 2> $ROOT/$OUT/rocprof_fr.err
cd $ROOT
python tools/rocpd_stats.py /tmp/prof_t/train_results.db 70 20 avgpool3x3_s1 2 > $OUT/train_kernel_stats.txt 2>&1
python tools/rocpd_stats.py /tmp/prof_s/sup_results.db 40 20 sup_grad_kernel > $OUT/sup_kernel_stats.txt 2>&1
python tools/rocpd_stats.py /tmp/prof_f/fr_results.db 40 > $OUT/fullres_kernel_stats.txt 2>&1
timeout 500 python tools/igemm_probe.py --rounds 3 --iters 10 2>&1 | grep -v amdgpu > $OUT/igemm_probe.txt
{ timeout 200 python tools/direct_dgrad_probe.py; timeout 200 python tools/heads_bwd_probe.py; } 2>&1 | grep -v amdgpu > $OUT/fused_backward_probe.txt
{ timeout 400 python tools/parity_sweep.py 30; timeout 400 python tools/parity_sweep_crf.py 60; timeout 600 python tools/parity_sweep_shapes.py 40; } 2>&1 | grep -v amdgpu > $OUT/parity_sweeps.txt
python tools/filter_trace.py 16 2>&1 | grep -v amdgpu > $OUT/filter_trace.txt
timeout 200 python tools/pylayers_route_cost.py 16 2>&1 | grep -v amdgpu > $OUT/pylayers_route.txt
timeout 300 python tools/skip_probe.py 16 2>&1 | grep -v amdgpu > $OUT/skip_probe.txt
timeout 300 python tools/grad_fidelity.py 16 2>&1 | grep -v amdgpu > $OUT/grad_fidelity_b16.txt
timeout 300 python tools/class_tiles_probe.py 16 2>&1 | grep -v amdgpu > $OUT/class_tiles_probe.txt
python tools/rocpd_timeline.py /tmp/prof_t/train_results.db avgpool3x3_s1 2 > $OUT/train_timeline.txt 2>&1
bash tools/gpu_trainf_profiles.sh $OUT > $OUT/trainf_profiles.log 2>&1
head -3 $OUT/train_kernel_stats.txt | cut -c1-200; tail -12 $OUT/parity_sweeps.txt; python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
print("value", d["value"], "ms", d["ms_per_step"], "fp32", d.get("value_fp32"), "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], d["roofline"]["traffic_source"].get("match"))
print({k: (v.get("value"), v.get("error")) for k, v in d.get("modes", {}).items()})
PY
