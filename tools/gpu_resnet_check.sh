export PYTHONPATH=$PWD
OUT=gpurun_out/r6_res; mkdir -p $OUT
run() { env "$@" timeout 600 python bench.py --mode train-f --backbone resnet101 --size 513 --batch 10 --steps 20 --warmup 6 --no-cpu-baseline 2>$OUT/bench.err | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$*', j['value'], j['ms_per_step'])"; }
for rep in 1 2; do
run A=1
run DSRG_RESNET_MERGED_BWD=0
done 2>&1 | tee $OUT/bench.log
run2() { env "$@" timeout 600 python bench.py --mode train --steps 30 --warmup 10 --no-cpu-baseline 2>$OUT/bench.err | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('train-s $*', j['value'], j['ms_per_step'])"; }
run2 A=1; run2 A=1
