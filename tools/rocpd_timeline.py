#!/usr/bin/env python
"""One steady-state step of a rocprofv3 kernel trace as a timeline: every dispatch in start order with its duration, the gap to
the previous dispatch's end, grid and workgroup size.  Usage: python tools/rocpd_timeline.py x_results.db marker [occ]
(a step = the span between two consecutive dispatches of the kernel whose name contains `marker`; the last full step is shown)."""
import re
import sqlite3
import sys


def main(path, marker, occ=1):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scol = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    dcol = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    name_col = "kernel_name" if "kernel_name" in scol else "display_name"
    gx = "d.grid_size_x" if "grid_size_x" in dcol else ("d.grid_x" if "grid_x" in dcol else "0")
    wx = "d.workgroup_size_x" if "workgroup_size_x" in dcol else ("d.workgroup_x" if "workgroup_x" in dcol else "1")
    rows = c.execute("select s.%s, d.start, d.end, %s, %s from %s d join %s s on d.kernel_id = s.id order by d.start" % (
        name_col, gx, wx, kd, ks)).fetchall()
    marks = [st for name, st, en, g, w in rows if marker in name]
    lo, hi = marks[-occ - 1], marks[-1]
    step = [r for r in rows if lo <= r[1] < hi]
    print("# %s: one step = %.3f ms wall, %d dispatches, %.3f ms of kernels, %.3f ms of gaps" % (
        path, (hi - lo) / 1e6, len(step), sum(r[2] - r[1] for r in step) / 1e6,
        ((hi - lo) - sum(r[2] - r[1] for r in step)) / 1e6))
    print("# columns: t_start_us  dur_us  gap_before_us  workgroups  name")
    prev = None
    for name, st, en, g, w in step:
        short = re.sub(r"^(void )?dsrg::(\(anonymous namespace\)::)?", "", name)
        short = re.sub(r"\(.*$", "", short)[:90]
        gap = (st - prev) / 1e3 if prev is not None else 0.0
        print("%10.1f %8.1f %7.1f %8d  %s" % ((st - lo) / 1e3, (en - st) / 1e3, gap, (g // max(w, 1)) if g else 0, short))
        prev = max(prev, en) if prev is not None else en


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1)
