export PYTHONPATH=$PWD
run2() { env "$@" timeout 600 python bench.py --mode train --steps 40 --warmup 10 --no-cpu-baseline --no-fp32 --no-modes --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('train-s $*', round(j['value'],1), round(j['ms_per_step'],4))"; }
for r in 1 2 3; do run2 A=1; run2 DSRG_MERGED_ORDER_ALL=1; done
