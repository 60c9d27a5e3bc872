export PYTHONPATH=$PWD
mkdir -p gpurun_out/r6_dma
for lib in libdsrg_hip.exp32.so libdsrg_hip.so libdsrg_hip.exp16.so libdsrg_hip.exp32.so libdsrg_hip.so libdsrg_hip.exp16.so; do
DSRG_LIB=$lib timeout 300 python tools/conv_direct_exp.py 2>&1 | grep -v amdgpu
done > gpurun_out/r6_dma/probe.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py -x -q -m gpu -k "direct or conv3x3 or backbone or vgg or train_step" > gpurun_out/r6_dma/pytest.log 2>&1; echo rc=$? >> gpurun_out/r6_dma/pytest.log
for lib in libdsrg_hip.exp32.so libdsrg_hip.so libdsrg_hip.exp16.so libdsrg_hip.exp32.so libdsrg_hip.so libdsrg_hip.exp16.so; do
DSRG_LIB=$lib timeout 300 python bench.py --mode train --steps 40 --warmup 10 --no-cpu-baseline --no-fp32 --no-modes --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib', round(j['value'],1), round(j['ms_per_step'],4))"
done > gpurun_out/r6_dma/bench.log 2>&1
cat gpurun_out/r6_dma/probe.log; tail -3 gpurun_out/r6_dma/pytest.log; cat gpurun_out/r6_dma/bench.log
