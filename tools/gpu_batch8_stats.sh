export PYTHONPATH=$PWD; ROOT=$PWD; OUT=gpurun_out/r6_blur; mkdir -p $OUT
python tools/crf_batch8_trace.py 10 2>&1 | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b8 -o b8 -- python $ROOT/tools/crf_batch8_trace.py 10 > /dev/null 2>&1
cd $ROOT
python tools/rocpd_stats.py /tmp/prof_b8/b8_results.db 40 > $OUT/batch8_kernel_stats.txt 2>&1
head -32 $OUT/batch8_kernel_stats.txt | cut -c1-160
