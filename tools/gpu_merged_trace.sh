export PYTHONPATH=$PWD; ROOT=$PWD; mkdir -p gpurun_out/r6_mtrace
cd /tmp && export TMPDIR=/tmp
for shape in "1024 256 65 1 1" "256 64 129 1 1" "256 1024 65 1 1" "256 256 65 3 2"; do
for e in A=1 DSRG_MERGED_W_FIRST=0 DSRG_IGEMM_MERGED_K1=0; do
rm -rf /tmp/prof_m
env $e timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_m -o m -- python $ROOT/tools/merged_bwd_trace.py $shape > /dev/null 2>&1
echo "== $shape $e"
python $ROOT/tools/rocpd_stats.py /tmp/prof_m/m_results.db 8 2>&1 | grep -E "igemm|reduce" | cut -c1-150
done; done > $ROOT/gpurun_out/r6_mtrace/trace.txt 2>&1
cat $ROOT/gpurun_out/r6_mtrace/trace.txt
