#!/bin/bash
# rocprofv3 kernel stats of the two train-f sub-records (VGG16-ASPP 321x321 batch 16; ResNet-101 513x513 batch 10)
# usage: bash tools/gpu_trainf_profiles.sh outdir
OUT=${1:-gpurun_out/trainf}
mkdir -p $OUT
export PYTHONPATH=$PWD
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_fv -o vgg -- python $ROOT/bench.py --mode train-f --backbone vgg16 --size 321 --batch 16 --steps 10 --warmup 5 --no-cpu-baseline > $ROOT/$OUT/bench_train_f_vgg_rocprof.json 2> $ROOT/$OUT/rocprof_vgg.err
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_fr -o r101 -- python $ROOT/bench.py --mode train-f --backbone resnet101 --size 513 --batch 10 --steps 6 --warmup 4 --no-cpu-baseline > $ROOT/$OUT/bench_train_f_r101_rocprof.json 2> $ROOT/$OUT/rocprof_r101.err
cd $ROOT
python tools/rocpd_stats.py /tmp/prof_fv/vgg_results.db 50 6 nll_loss2d_forward > $OUT/train_f_vgg16_kernel_stats.txt 2>&1
python tools/rocpd_stats.py /tmp/prof_fr/r101_results.db 70 4 max_pool_backward > $OUT/train_f_resnet101_513_kernel_stats.txt 2>&1
head -4 $OUT/train_f_vgg16_kernel_stats.txt | cut -c1-160; head -30 $OUT/train_f_resnet101_513_kernel_stats.txt | cut -c1-160
