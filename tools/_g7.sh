mkdir -p gpurun_out/g7
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "crf" 2>&1 | tail -15 | tee gpurun_out/g7/pytest_crf.txt
timeout 600 python bench.py --mode crf-fullres --steps 20 --warmup 5 > gpurun_out/g7/bench_fullres.json 2> gpurun_out/g7/bench_fullres.err
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/g7/bench_fullres.json"))
    print({k: d.get(k) for k in ("value","ms_per_step","images_per_s_four_in_flight","images_per_s_batch8","ms_per_image_batch8","images_per_s_batch8_two_in_flight","batch8_splat","max_abs_dq_vs_oracle")})
    print(d["roofline"]["frac"], d["roofline"]["us_per_launch"])
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/g7/bench_fullres.err").read()[-3000:])
PY
