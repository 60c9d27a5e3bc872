mkdir -p gpurun_out/g9
export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "pylayers or protocol or unused or glue" 2>&1 | tail -12 | tee gpurun_out/g9/pytest_layers.txt
timeout 300 python tools/pylayers_route_cost.py 16 2>&1 | grep -v amdgpu | tee gpurun_out/g9/route_cost.txt
DSRG_PYLAYERS_PIN=0 timeout 300 python tools/pylayers_route_cost.py 16 2>&1 | grep -v amdgpu | tail -1 | tee gpurun_out/g9/route_cost_nopin.txt
DSRG_PYLAYERS_TRUST=1 timeout 300 python tools/pylayers_route_cost.py 16 2>&1 | grep -v amdgpu | tail -1 | tee gpurun_out/g9/route_cost_trust.txt
DSRG_PYLAYERS_EXACT=1 timeout 300 python tools/pylayers_route_cost.py 16 2>&1 | grep -v amdgpu | tail -1 | tee gpurun_out/g9/route_cost_exact.txt
