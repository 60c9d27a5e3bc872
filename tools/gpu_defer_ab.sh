export PYTHONPATH=$PWD
OUT=gpurun_out/r6_defer; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_parity.py -x -q -m gpu -k "deferred or config3 or sgd_pack or ddp_rccl_single or bias or relu_bwd or maxpool" > $OUT/pytest.log 2>&1; echo rc=$? >> $OUT/pytest.log; tail -4 $OUT/pytest.log
run2() { env "$@" timeout 600 python bench.py --mode train --steps 40 --warmup 10 --no-cpu-baseline --no-fp32 --no-modes --no-profile 2>$OUT/bench.err | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('train-s $*', round(j['value'],1), round(j['ms_per_step'],4))"; }
for r in 1 2 3; do run2 DSRG_DEFER_REDUCTIONS=1; run2 DSRG_DEFER_REDUCTIONS=0; done
run3() { env "$@" timeout 600 python bench.py --mode train-f --backbone vgg16 --size 321 --batch 16 --steps 20 --warmup 6 --no-cpu-baseline 2>$OUT/bench.err | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('train-f $*', round(j['value'],1), round(j['ms_per_step'],4))"; }
run3 DSRG_DEFER_REDUCTIONS=1; run3 DSRG_DEFER_REDUCTIONS=0
