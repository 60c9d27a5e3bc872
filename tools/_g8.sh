mkdir -p gpurun_out/g8
export PYTHONPATH=$PWD
ROOT=$PWD
for nb in 1 4 8; do timeout 200 python tools/crf_batch_probe.py $nb 10 2>&1 | grep -v amdgpu | tee -a gpurun_out/g8/probe.txt; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o fr -- python $ROOT/tools/crf_batch_probe.py 8 10 > /dev/null 2> $ROOT/gpurun_out/g8/rocprof.err
cd $ROOT
python tools/rocpd_stats.py /tmp/prof/fr_results.db 40 > gpurun_out/g8/batch8_kernel_stats.txt 2>&1
head -40 gpurun_out/g8/batch8_kernel_stats.txt | cut -c1-170
