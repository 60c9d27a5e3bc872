mkdir -p gpurun_out/g11
export PYTHONPATH=$PWD
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee gpurun_out/g11/pytest_all.txt
timeout 1500 python bench.py > gpurun_out/g11/bench_default.json 2> gpurun_out/g11/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/g11/bench_default.json"))
print("value", d["value"], "ms", d["ms_per_step"], "fp32", d.get("value_fp32"), "frac", d["roofline"]["frac"], d["roofline"]["us_per_launch"])
print({k: v for k, v in d["legs"].items() if k.startswith("grad") or k.startswith("loss_gap") or k.endswith("error")})
print({k: (v.get("value"), v.get("error")) for k, v in d.get("modes", {}).items()})
print({k: d["modes"]["crf_fullres"].get(k) for k in ("images_per_s_batch8","images_per_s_four_in_flight","ms_per_image_batch8")})
print("weights_equal", d.get("weights_equal_across_ranks"), d.get("weights_checksum_rank0"))
PY
