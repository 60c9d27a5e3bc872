export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_igemm.py tests/test_gpu_trainer.py -q 2>&1 | tail -4
for v in 1 1; do
timeout 300 python bench.py --steps 20 --warmup 8 --no-fp32 --no-cpu-baseline --no-modes --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('now', d['value'], d['ms_per_step'], d['losses'])"
done
DSRG_IGEMM_VARIANT=6 DSRG_MERGED_BWD=0 timeout 300 python bench.py --steps 20 --warmup 8 --no-fp32 --no-cpu-baseline --no-modes --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round-4 launches', d['value'], d['ms_per_step'], d['losses'])"
