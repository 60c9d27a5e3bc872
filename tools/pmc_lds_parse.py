#!/usr/bin/env python
"""LDS-side counters of the hot-path kernels from one `rocprofv3 --pmc ...` pass (counter_collection CSV) ->
profiles/rNN_lds_counters.json.

  python tools/pmc_lds_parse.py COUNTER_CSV OUT_JSON [commit]

Per kernel (mean per launch, summed over the chip as rocprofv3 reports them): every counter of the pass, plus for the
mean-field filter the derived figures DESIGN.md quotes: LDS-array busy fraction = SQ_LDS_IDX_ACTIVE / (duration x clock
x 256 CUs), conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (MI355X_MICROARCH.md: BANK_CONFLICT = extra
cycles, IDX_ACTIVE = all LDS-array cycles).  Durations come from the same CSV (Start/End timestamps)."""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd.provenance import all_sources_sha256  # noqa: E402

CUS = 256
XCDS = 8


def main(path, out, commit=""):
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        did = r.get("Dispatch_Id")
        if (k, did) not in seen and r.get("Start_Timestamp"):
            seen.add((k, did))
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    res = {"commit": commit, "sources_sha256": all_sources_sha256(), "source": path, "note": "means per launch; SQ counters are summed over all SEs/CUs; durations under "
           "counter collection are longer than un-profiled ones (serialised dispatches, lower clock)", "kernels": {}}
    for k, cs in acc.items():
        if "dsrg::" not in k:
            continue
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        e["launches"] = max(len(v) for v in cs.values())
        if dur[k]:
            e["avg_duration_us_under_pmc"] = sum(dur[k]) / len(dur[k]) / 1e3
        if "SQ_LDS_IDX_ACTIVE" in e and e["SQ_LDS_IDX_ACTIVE"] > 0:
            e["lds_conflict_share"] = e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"]
            if "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"] > 0:
                # GRBM_GUI_ACTIVE comes summed over the 8 XCDs: cycles of ONE clock domain = value / 8
                e["lds_array_busy_frac"] = e["SQ_LDS_IDX_ACTIVE"] / (e["GRBM_GUI_ACTIVE"] / XCDS * CUS)
            if e.get("avg_duration_us_under_pmc"):
                e["lds_array_busy_frac_at_2.4GHz"] = e["SQ_LDS_IDX_ACTIVE"] / (CUS * e["avg_duration_us_under_pmc"] * 1e-6 * 2.4e9)
        res["kernels"][k] = e
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, e in sorted(res["kernels"].items()):
        if "mf_" in k or "srg" in k:
            print(k, json.dumps(e, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
