export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_igemm.py tests/test_gpu_trainer.py tests/test_inference.py -q 2>&1 | tail -4
for v in 0 1 0 1; do
DSRG_PREPACK=$v timeout 300 python bench.py --steps 20 --warmup 8 --no-fp32 --no-cpu-baseline --no-modes --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prepack $v', d['value'], d['ms_per_step'], d['losses'])"
done
