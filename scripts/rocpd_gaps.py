#!/usr/bin/env python3
"""Where the GPU sits idle in a rocprofv3 kernel trace (rocpd sqlite database): the dispatches in
start order, the time between the end of everything before a dispatch and its start, summed per
(kernel before -> kernel after).  usage: rocpd_gaps.py results.db [min_gap_us [from to]] -- from / to: the part of the
trace to look at, as fractions of its span (0.4 0.8 = the steady state of a bench run)."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 2e3
disp = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
sym = [r[0] for r in cur.execute("select name from sqlite_master where type='table' and name like 'rocpd_info_kernel_symbol%'")][0]
rows = list(cur.execute("select d.start, d.end, s.kernel_name from %s d join %s s on d.kernel_id = s.id order by d.start" % (disp, sym)))
if len(sys.argv) > 4:
    t0, t1 = rows[0][0], rows[-1][1]
    lo, hi = t0 + float(sys.argv[3]) * (t1 - t0), t0 + float(sys.argv[4]) * (t1 - t0)
    rows = [r for r in rows if lo <= r[0] <= hi]
def short(n):
    n = n.split("(")[0]
    return n.replace("bwtc_hip::", "").replace("void ", "")[:44]
busy_end, prev = rows[0][1], rows[0][2]
gaps = defaultdict(lambda: [0, 0.0])
busy = 0.0
for st, en, name in rows[1:]:
    if st > busy_end:
        g = st - busy_end
        if g >= min_gap:
            k = (short(prev), short(name))
            gaps[k][0] += 1
            gaps[k][1] += g
        busy_end_new = en
    if en > busy_end:
        busy += en - max(st, busy_end)
        busy_end = en
        prev = name
span = rows[-1][1] - rows[0][0]
idle = span - busy - (rows[0][1] - rows[0][0])
print("span %.1f ms, busy %.1f ms, idle %.1f ms (%.1f %%)" % (span / 1e6, busy / 1e6, idle / 1e6, 100.0 * idle / span))
for (a, b), (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%8.3f ms in %4d gaps (avg %7.1f us)  %s -> %s" % (t / 1e6, n, t / n / 1e3, a, b))
