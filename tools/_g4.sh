mkdir -p gpurun_out/g4
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_igemm.py -q 2>&1 | tail -8 > gpurun_out/g4/pytest_igemm.txt
cat gpurun_out/g4/pytest_igemm.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "pool" 2>&1 | tail -8 | tee gpurun_out/g4/pytest_pool.txt
timeout 300 python tools/skip_probe.py 16 2>&1 | grep -v amdgpu | tee gpurun_out/g4/skip_probe.txt
for v in 6 3 6 3; do
DSRG_IGEMM_VARIANT=$v timeout 300 python bench.py --steps 20 --warmup 8 --no-fp32 --no-cpu-baseline --no-modes --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $v', d['value'], d['ms_per_step'], d['losses'])" | tee -a gpurun_out/g4/ab_skip.txt
done
for lr in 2e-5 4e-6; do
timeout 300 python tools/overfit_probe.py 300 8 --init kaiming --dropout 0 --lr $lr 2>&1 | grep -v amdgpu > gpurun_out/g4/overfit_kaiming_lr$lr.txt; tail -2 gpurun_out/g4/overfit_kaiming_lr$lr.txt
done
