export PYTHONPATH=$PWD
OUT=gpurun_out/r6_blur; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "crf or large or fullres or batch" > $OUT/pytest.log 2>&1; echo rc=$? >> $OUT/pytest.log; tail -2 $OUT/pytest.log
for lib in libdsrg_hip.base.so libdsrg_hip.so libdsrg_hip.base.so libdsrg_hip.so; do
DSRG_LIB=$lib timeout 300 python bench.py --mode crf-fullres --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib', 'single ms', round(j['ms_per_step'],4), 'b8 ms/img', round(j['ms_per_image_batch8'],4), 'b8x2', round(j['images_per_s_batch8_two_in_flight'],1), 'splat b8 us', round(j['batch8_splat']['us_per_launch'],1))"
done | tee $OUT/ab.log
