mkdir -p gpurun_out/g18
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_igemm.py -q -x 2>&1 | tail -4
for v in 0 1 0 1; do
DSRG_MERGED_KS=$v timeout 300 python bench.py --steps 20 --warmup 8 --no-fp32 --no-cpu-baseline --no-modes --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merged_ks $v', d['value'], d['ms_per_step'], d['losses'])" | tee -a gpurun_out/g18/ab_merged_ks.txt
done
DSRG_MERGED_BWD=0 timeout 300 python bench.py --steps 20 --warmup 8 --no-fp32 --no-cpu-baseline --no-modes --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merged_bwd 0', d['value'], d['ms_per_step'], d['losses'])" | tee -a gpurun_out/g18/ab_merged_ks.txt
