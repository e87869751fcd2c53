#!/usr/bin/env python3
"""Per-block view of a rocprofv3 kernel-stats CSV (scripts/rocpd_stats.py): calls and ms per block."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
blocks = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
pat = sys.argv[3] if len(sys.argv) > 3 else ""
tot = 0.0
for r in rows:
    ms = float(r["TotalDurationNs"]) / 1e6 / blocks
    tot += ms
    if pat and pat not in r["Name"]:
        continue
    if ms < 0.02:
        continue
    print(f"{r['Name'][:84]:84s} {float(r['Calls'])/blocks:7.1f} {ms:8.3f} ms  avg {float(r['AverageNs'])/1e3:9.1f} us")
print(f"total per block {tot:.2f} ms")
