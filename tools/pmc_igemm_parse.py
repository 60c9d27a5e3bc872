#!/usr/bin/env python
"""per-kernel means of a rocprofv3 counter_collection.csv (one row per dispatch and counter): usage pmc_igemm_parse.py file.csv"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
agg = defaultdict(lambda: defaultdict(list))
for r in rows:
    agg[(r["Kernel_Name"][:60], r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, grid), cs in sorted(agg.items()):
    if "igemm" not in k:
        continue
    print(k, "grid", grid, "launches", max(len(v) for v in cs.values()))
    for c, v in sorted(cs.items()):
        print("    %-28s %.4g" % (c, sum(v) / len(v)))
