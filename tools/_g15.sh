mkdir -p gpurun_out/g15
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_igemm.py -q 2>&1 | tail -4
timeout 300 python tools/skip_probe.py 16 2>&1 | grep -v amdgpu | grep "weight" | tee gpurun_out/g15/skip_probe.txt
timeout 400 python tools/igemm_probe.py --rounds 3 --iters 10 2>&1 | grep -v amdgpu | tee gpurun_out/g15/igemm_probe.txt | tail -12
for v in 6 3 6 3; do
DSRG_IGEMM_VARIANT=$v timeout 300 python bench.py --steps 20 --warmup 8 --no-fp32 --no-cpu-baseline --no-modes --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $v', d['value'], d['ms_per_step'], d['losses'])" | tee -a gpurun_out/g15/ab_all.txt
done
