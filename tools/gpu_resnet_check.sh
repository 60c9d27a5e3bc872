export PYTHONPATH=$PWD
OUT=gpurun_out/r6_res; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py -x -q -m gpu -k "resnet or aspp or retrain or train_f" > $OUT/pytest.log 2>&1; echo rc=$? >> $OUT/pytest.log
tail -3 $OUT/pytest.log
run() { env "$@" timeout 600 python bench.py --mode train-f --backbone resnet101 --size 513 --batch 10 --steps 20 --warmup 6 --no-cpu-baseline 2>$OUT/bench.err | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$*', j['value'], j['ms_per_step'])"; }
run A=1; run A=1
