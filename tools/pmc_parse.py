#!/usr/bin/env python
"""Turn the counter_collection CSVs of two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into
profiles/pmc_traffic.json: HBM bytes per launch for every hot-path kernel, corrected by the factor
measured on the calibration launch (known byte count) as MI355X_MICROARCH.md §HBM prescribes.
Usage: python tools/pmc_parse.py FETCH_CSV WRITE_CSV OUT_JSON [commit]   (the commit the counters were collected at is
stamped into the file: re-collect when a kernel changes)"""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd.provenance import all_sources_sha256  # noqa: E402

CALIB_KERNEL = "softmax_fwd_kernel"
CALIB_BYTES = 2048 * 21 * 41 * 41 * 4


def per_kernel(path, counter):
    acc = defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        acc[(row["Kernel_Name"], int(row["Grid_Size"]))].append(float(row["Counter_Value"]))
    return acc


def main(fetch_csv, write_csv, out, commit=""):
    res = {"commit": commit, "sources_sha256": all_sources_sha256(), "units": "bytes per launch; raw counters are KiB (FETCH_SIZE/WRITE_SIZE), corrected by the "
                    "calibration factor of a %d-byte softmax_fwd_kernel launch" % CALIB_BYTES}
    factors = {}
    tables = {}
    for name, path in (("FETCH_SIZE", fetch_csv), ("WRITE_SIZE", write_csv)):
        t = per_kernel(path, name)
        tables[name] = t
        cal = [v for (k, g), vals in t.items() if CALIB_KERNEL in k and g > 1000000 for v in vals]
        raw = sum(cal) / len(cal) * 1024.0
        factors[name] = CALIB_BYTES / raw
        res["calibration_%s" % name] = {"raw_bytes": raw, "true_bytes": CALIB_BYTES, "factor": factors[name]}
    kernels = {}
    for name, t in tables.items():
        for (k, g), vals in t.items():
            if CALIB_KERNEL in k and g > 1000000:
                continue
            short = k.replace("(anonymous namespace)::", "").split("(")[0]
            e = kernels.setdefault(short, {})
            e[name + "_raw_bytes"] = sum(vals) / len(vals) * 1024.0
            e[name + "_bytes"] = e[name + "_raw_bytes"] * factors[name]
            e["launches_" + name] = len(vals)
    for k, e in kernels.items():
        e["hbm_bytes_per_launch"] = e.get("FETCH_SIZE_bytes", 0.0) + e.get("WRITE_SIZE_bytes", 0.0)
    res["kernels"] = kernels
    # the inference loop's instantiation, not the one-plane launch of the lattice build's norm pass: the one launched most
    filt = sorted((e for k, e in kernels.items() if "mf_filter_kernel" in k), key=lambda e: -e.get("launches_FETCH_SIZE", 0))
    if filt:
        res["mf_filter_kernel_bytes_per_launch"] = filt[0]["hbm_bytes_per_launch"]
    for name in ("lg_blur2_kernel", "lg_splat2_kernel", "lg_slice_update_kernel"):
        sel = [e for k, e in kernels.items() if name in k]
        if sel:
            res[name + "_bytes_per_launch"] = sel[0]["hbm_bytes_per_launch"]
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True)[:3000])


if __name__ == "__main__":
    main(*sys.argv[1:5])
