#!/usr/bin/env python
"""what the host was doing while the GPU sat idle in front of a kernel: for the last dispatch of the kernel whose name contains `marker`
in a rocprofv3 --hip-trace --kernel-trace database, the HIP API calls that overlap [end of the previous kernel, start of this one]
usage: python tools/rocpd_gap.py x_results.db marker"""
import sqlite3, sys


def main(path, marker):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    reg = [t for t in tabs if t.startswith("rocpd_region")][0]
    st = [t for t in tabs if t.startswith("rocpd_string")][0]
    scol = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "kernel_name" if "kernel_name" in scol else "display_name"
    rows = c.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, kd, ks)).fetchall()
    idx = [i for i, r in enumerate(rows) if marker in r[0]][-2]
    prev_end = max(r[2] for r in rows[max(0, idx - 40):idx])
    start = rows[idx][1]
    print("gap in front of %s: %.1f us" % (rows[idx][0][:60], (start - prev_end) / 1e3))
    for i in range(max(0, idx - 6), idx + 1):
        print("   kernel %-70s start %+9.1f us  dur %7.1f" % (rows[i][0][:70], (rows[i][1] - start) / 1e3, (rows[i][2] - rows[i][1]) / 1e3))
    api = c.execute("select s.string, r.start, r.end from %s r join %s s on r.name_id = s.id where r.end > ? and r.start < ? order by r.start" % (reg, st),
                    (prev_end - 300000, start + 20000)).fetchall()
    for name, a, b in api:
        print("   host   %-40s %+9.1f .. %+9.1f us (%.1f)" % (name[:40], (a - start) / 1e3, (b - start) / 1e3, (b - a) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
