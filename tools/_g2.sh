mkdir -p gpurun_out/g2
export PYTHONPATH=$PWD
timeout 300 python tools/grad_fidelity.py 2 2>&1 | grep -v amdgpu > gpurun_out/g2/fidelity_b2.txt
timeout 400 python tools/overfit_probe.py 300 8 2>&1 | grep -v amdgpu > gpurun_out/g2/overfit_default.txt
timeout 300 python tools/overfit_probe.py 300 8 --init kaiming --dropout 0 2>&1 | grep -v amdgpu > gpurun_out/g2/overfit_kaiming.txt
timeout 600 python -m pytest tests/test_gpu_trainer.py -q -x -k "launcher or rccl" 2>&1 | tail -5 > gpurun_out/g2/pytest_rccl.txt
tail -3 gpurun_out/g2/fidelity_b2.txt | cut -c1-400; tail -3 gpurun_out/g2/overfit_default.txt; tail -2 gpurun_out/g2/overfit_kaiming.txt; cat gpurun_out/g2/pytest_rccl.txt
