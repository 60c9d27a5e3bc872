#!/bin/bash
# ResNet-101 train-f: tests of the shortcut-in-the-store path, then the train-f-resnet bench with and without it
export PYTHONPATH=$PWD
OUT=gpurun_out/r6_res; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_igemm.py -x -q -m gpu -k "residual or resnet or shortcut or igemm" > $OUT/pytest.log 2>&1; echo rc=$? >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for v in 1 0 1 0; do
DSRG_RESNET_FUSE_RES=$v timeout 600 python bench.py --mode train-f --backbone resnet101 --size 513 --batch 10 --steps 10 --warmup 4 --no-cpu-baseline 2>$OUT/bench_$v.err | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('fuse_res=$v', j['value'], j['ms_per_step'])"
done 2>&1 | tee $OUT/bench.log
