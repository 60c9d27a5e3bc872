export PYTHONPATH=$PWD; ROOT=$PWD; OUT=gpurun_out/r6_trainf2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_fr -o r101 -- python $ROOT/bench.py --mode train-f --backbone resnet101 --size 513 --batch 10 --steps 6 --warmup 4 --no-cpu-baseline > $ROOT/$OUT/bench_train_f_r101_rocprof.json 2> $ROOT/$OUT/rocprof_r101.err
cd $ROOT
python tools/rocpd_stats.py /tmp/prof_fr/r101_results.db 70 4 max_pool_backward > $OUT/train_f_resnet101_513_kernel_stats.txt 2>&1
head -50 $OUT/train_f_resnet101_513_kernel_stats.txt | cut -c1-180
