#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 run that wrote a rocpd sqlite database (ROCm 7.2's default
output): name, calls, total ms, average us, share -- the table `--stats` prints as CSV in older
versions.  usage: rocpd_stats.py results.db [out.csv]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
disp = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
sym = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_info_kernel_symbol%'")][0]
rows = list(cur.execute("select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
                        "max(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 3 desc" % (disp, sym)))
tot = sum(r[2] for r in rows) or 1
out = csv.writer(open(sys.argv[2], "w", newline="")) if len(sys.argv) > 2 else None
if out:
    out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for r in rows:
    if out:
        out.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.3f" % (100.0 * r[2] / tot), r[4], r[5]])
    else:
        print("%-100s %6d %10.3f ms %10.1f us %5.1f%%" % (r[0][:100], r[1], r[2] / 1e6, r[3] / 1e3, 100.0 * r[2] / tot))
