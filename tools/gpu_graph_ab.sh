export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_inference.py -x -q -m gpu 2>&1 | tail -15
for v in 1 0 1 0; do DSRG_TEST_MS_GRAPH=$v timeout 300 python bench.py --mode test-ms --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('graph=$v', round(j['value'],1), round(j['ms_per_step'],3), round(j['forwards_zoom_softmax_ms'],3), round(j['crf_argmax_ms'],3))"; done
