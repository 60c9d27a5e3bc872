mkdir -p gpurun_out/g1
export PYTHONPATH=$PWD
timeout 300 python tools/grad_fidelity.py 2 2>&1 | grep -v amdgpu > gpurun_out/g1/fidelity_b2.txt
timeout 400 python tools/overfit_probe.py 300 8 2>&1 | grep -v amdgpu > gpurun_out/g1/overfit_default.txt
timeout 300 python tools/overfit_probe.py 300 8 --init kaiming --dropout 0 2>&1 | grep -v amdgpu > gpurun_out/g1/overfit_kaiming.txt
timeout 300 python bench.py --steps 20 --warmup 8 --no-fp32 --no-cpu-baseline --no-modes > gpurun_out/g1/bench_quick.json 2>gpurun_out/g1/bench_quick.err
tail -5 gpurun_out/g1/fidelity_b2.txt; tail -3 gpurun_out/g1/overfit_default.txt; tail -2 gpurun_out/g1/overfit_kaiming.txt; cut -c1-300 gpurun_out/g1/bench_quick.json
