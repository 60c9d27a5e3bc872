mkdir -p gpurun_out/g14
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_igemm.py -q 2>&1 | tail -6 | tee gpurun_out/g14/pytest_igemm.txt
timeout 300 python tools/skip_probe.py 16 2>&1 | grep -v amdgpu | tee gpurun_out/g14/skip_probe.txt
for v in 0 1 0 1; do
DSRG_WGRAD_COMPACT=$v timeout 300 python bench.py --steps 20 --warmup 8 --no-fp32 --no-cpu-baseline --no-modes --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('compact $v', d['value'], d['ms_per_step'], d['losses'])" | tee -a gpurun_out/g14/ab_compact.txt
done
timeout 600 python -m pytest tests/test_gpu_trainer.py -q -k "config3 or dropout or float32_gradient or prepared" 2>&1 | tail -4
