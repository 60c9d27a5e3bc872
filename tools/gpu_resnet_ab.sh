#!/bin/bash
# ResNet-101 train-f: tests of the shortcut-in-the-store / merged-backward path, then the train-f ResNet bench with the switches
export PYTHONPATH=$PWD
OUT=gpurun_out/r6_res; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_igemm.py tests/test_gpu_trainer.py -x -q -m gpu -k "residual or resnet or shortcut or igemm or pack" > $OUT/pytest.log 2>&1; echo rc=$? >> $OUT/pytest.log
tail -5 $OUT/pytest.log
run() { env "$@" timeout 600 python bench.py --mode train-f --backbone resnet101 --size 513 --batch 10 --steps 20 --warmup 6 --no-cpu-baseline 2>$OUT/bench.err | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$*', j['value'], j['ms_per_step'])"; }
for rep in 1 2 3; do
run A=1
run DSRG_RESNET_MERGED_BWD=0
run DSRG_IGEMM_MERGED_K1=0
done 2>&1 | tee $OUT/bench.log
