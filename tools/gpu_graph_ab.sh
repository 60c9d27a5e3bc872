export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_inference.py -x -q -m gpu 2>&1 | tail -3
for cfg in "3 1" "4 1" "3 2" "4 2" "3 1"; do set -- $cfg; DSRG_TEST_MS_INFLIGHT=$1 DSRG_TEST_MS_BATCH=$2 timeout 300 python bench.py --mode test-ms --steps 20 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('in_flight=$1 batch=$2', round(j['value'],1), 'pipelined', round(j['images_per_s_crfs_in_flight'],1))"; done
