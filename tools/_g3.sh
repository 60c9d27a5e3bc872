mkdir -p gpurun_out/g3
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_igemm.py -q -x 2>&1 | tail -8 > gpurun_out/g3/pytest_igemm.txt
cat gpurun_out/g3/pytest_igemm.txt
for v in 6 3 6 3; do
DSRG_IGEMM_VARIANT=$v timeout 300 python bench.py --steps 20 --warmup 8 --no-fp32 --no-cpu-baseline --no-modes --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $v', d['value'], d['ms_per_step'], d['losses'])" | tee -a gpurun_out/g3/ab_skip.txt
done
timeout 300 python tools/grad_fidelity.py 2 2>&1 | grep -v amdgpu > gpurun_out/g3/fidelity_b2.txt
timeout 300 python tools/grad_fidelity.py 16 2>&1 | grep -v amdgpu > gpurun_out/g3/fidelity_b16.txt
timeout 300 python tools/overfit_probe.py 300 8 --init kaiming --dropout 0 2>&1 | grep -v amdgpu > gpurun_out/g3/overfit_kaiming.txt
tail -2 gpurun_out/g3/fidelity_b2.txt | cut -c1-400; tail -2 gpurun_out/g3/fidelity_b16.txt| cut -c1-400; tail -2 gpurun_out/g3/overfit_kaiming.txt
