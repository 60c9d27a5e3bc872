#!/usr/bin/env python
"""LDS counters of a rocprofv3 --pmc pass summed per kernel name: bank-conflict cycles against LDS-active cycles
usage: pmc_lds_by_kernel.py counter_collection.csv"""
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name") or r.get("kernel_name")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k] += 1
rows = []
for k, c in agg.items():
    act = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    rows.append((act, k, c))
rows.sort(reverse=True)
print("%-14s %-14s %-8s %-14s %-12s  %s" % ("lds_active", "bank_conflict", "confl %", "insts_lds", "busy_cycles", "kernel"))
for act, k, c in rows[:14]:
    bc = c.get("SQ_LDS_BANK_CONFLICT", 0.0)
    print("%-14.4g %-14.4g %-8.1f %-14.4g %-12.4g  %s" % (act, bc, 100.0 * bc / act if act else 0.0, c.get("SQ_INSTS_LDS", 0.0), c.get("SQ_BUSY_CYCLES", 0.0), k[:90]))
