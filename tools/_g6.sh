mkdir -p gpurun_out/g6
export PYTHONPATH=$PWD
for v in 0 1 0 1; do
DSRG_WGRAD_SIDE=$v timeout 300 python bench.py --steps 20 --warmup 8 --no-fp32 --no-cpu-baseline --no-modes --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad_side $v', d['value'], d['ms_per_step'], d['losses'])" | tee -a gpurun_out/g6/ab_side.txt
done
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee gpurun_out/g6/pytest_all.txt
bash tools/gpu_ab_lib.sh gpurun_out/g6/ab_head libdsrg_hip.so libdsrg_hip.exp16.so > gpurun_out/g6/ab_head.log 2>&1
tail -12 gpurun_out/g6/ab_head.log
