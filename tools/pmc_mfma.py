#!/usr/bin/env python
"""MFMA utilisation of the train step from one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES` pass.

  python tools/pmc_mfma.py COUNTER_CSV [steps] > profiles/rNN_mfma_utilisation.txt

A step is delimited by the dispatches of avgpool3x3_s1 (two per step).  SQ_VALU_MFMA_BUSY_CYCLES is summed over the
SIMDs of the chip (MI355X_MICROARCH.md: = 32 x the number of 32x32x16 bf16 MFMAs), so
utilisation = busy cycles / (kernel duration x clock x 1024 SIMDs); the clock is taken as 2.4 GHz (peak), which makes
the figure a lower bound when the chip runs below it."""
import csv
import sys
from collections import defaultdict

CLOCK_GHZ, SIMDS = 2.4, 1024


def main(path, steps=3):
    rows = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES":
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    marks = [s for s, e, n, v in rows if "avgpool3x3_s1" in n]
    if len(marks) < 2 * steps + 1:
        raise SystemExit("only %d marker dispatches" % len(marks))
    lo, hi = marks[-2 * steps - 1], marks[-1]
    sel = [r for r in rows if lo <= r[0] < hi]
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for s, e, n, v in sel:
        a = agg[n[:110]]
        a[0] += 1
        a[1] += (e - s)
        a[2] += v
    tot_ns = sum(a[1] for a in agg.values())
    tot_busy = sum(a[2] for a in agg.values())
    print("# %s: last %d train steps, %d dispatches per step, %.3f ms of kernel time per step (serialised by the counter pass)" % (
        path, steps, len(sel) // steps, tot_ns / steps / 1e6))
    print("# MFMA utilisation of the whole step: %.1f %% of the MFMA issue cycles at %.1f GHz (busy %.3e cycles per step)" % (
        100.0 * tot_busy / (tot_ns * CLOCK_GHZ * SIMDS), CLOCK_GHZ, tot_busy / steps))
    print("# columns: calls/step  us/step  mfma_util%  share_of_mfma_cycles%  kernel")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][2])[:24]:
        print("%6.1f %10.1f %8.1f %8.1f   %s" % (a[0] / steps, a[1] / steps / 1e3, 100.0 * a[2] / (a[1] * CLOCK_GHZ * SIMDS),
                                               100.0 * a[2] / max(tot_busy, 1.0), n))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
