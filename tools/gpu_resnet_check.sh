export PYTHONPATH=$PWD
OUT=gpurun_out/r6_res; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_igemm.py tests/test_gpu_trainer.py -x -q -m gpu -k "residual or resnet or shortcut or igemm or pack or retrain or train_f" > $OUT/pytest.log 2>&1; echo rc=$? >> $OUT/pytest.log
tail -5 $OUT/pytest.log

