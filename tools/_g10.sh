mkdir -p gpurun_out/g10
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "pylayers or protocol or unused or glue or unsupported or beyond or losses or fused" 2>&1 | tail -12 | tee gpurun_out/g10/pytest_layers.txt
timeout 300 python tools/pylayers_route_cost.py 16 2>&1 | grep -v amdgpu | tee gpurun_out/g10/route_cost.txt
DSRG_PYLAYERS_TRUST=1 timeout 300 python tools/pylayers_route_cost.py 16 2>&1 | grep -v amdgpu | tail -1 | tee gpurun_out/g10/route_cost_trust.txt
DSRG_PYLAYERS_EXACT=1 timeout 300 python tools/pylayers_route_cost.py 16 2>&1 | grep -v amdgpu | tail -1 | tee gpurun_out/g10/route_cost_exact.txt
