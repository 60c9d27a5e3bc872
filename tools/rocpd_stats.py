#!/usr/bin/env python
"""Per-kernel summary (calls, total, avg, min, max, share) of a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes DIR/NAME_results.db).
Usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db [top [steps marker]] > profiles/rNN_x_kernel_stats.txt
With `steps marker` only the last `steps` steady-state steps are summarised (per step): a step is the span between two
consecutive dispatches of the kernel whose name contains `marker` (one that runs once per step, e.g. sup_grad_kernel;
a fifth argument gives the number of dispatches of the marker per step when it is not 1)."""
import sqlite3
import sys


def main(path, top=40, steps=0, marker=None, occ=1):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    scol = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "kernel_name" if "kernel_name" in scol else "display_name"
    rows = c.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id" % (name_col, kd, ks)).fetchall()
    div = 1
    if steps and marker:
        marks = sorted(st for name, st, en in rows if marker in name)
        if len(marks) < steps * occ + 1:
            raise SystemExit("only %d dispatches of %s" % (len(marks), marker))
        lo, hi = marks[-steps * occ - 1], marks[-1]
        rows = [r for r in rows if lo <= r[1] < hi]
        div = steps
        print("# last %d steps between dispatches of *%s*: wall %.3f ms per step; figures below are PER STEP" % (
            steps, marker, (hi - lo) / steps / 1e6))
    agg = {}
    for name, st, en in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        dur = en - st
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values()) or 1
    print("# %s — %d dispatches, %.3f ms of kernel time (columns: calls total_us avg_us min_us max_us pct name)" % (
        path, len(rows) // div, total / div / 1e6))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%7.1f %12.1f %10.2f %10.2f %10.2f %6.2f%%  %s" % (a[0] / div, a[1] / div / 1e3, a[1] / a[0] / 1e3, a[2] / 1e3,
                                                              a[3] / 1e3, 100.0 * a[1] / total, name[:150]))
    return cols


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40,
         int(sys.argv[3]) if len(sys.argv) > 4 else 0, sys.argv[4] if len(sys.argv) > 4 else None,
         int(sys.argv[5]) if len(sys.argv) > 5 else 1)
