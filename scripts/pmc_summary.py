#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) into a per-kernel HBM-traffic summary.

gfx950 correction: FETCH_SIZE counts 128-B requests at 64 B, i.e. reports half the bytes
of a streaming read (calibrated here on k_radix_hist<u64>, whose read is exactly 8 B * n,
and on the 256 MiB device copy); WRITE_SIZE is exact for streaming stores.  Both counters
are in KiB.  Usage: pmc_summary.py <fetch_dir> <write_dir> <out.json> [workload note]"""
import collections
import csv
import glob
import json
import sys


def load(d):
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    out = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return out


def main():
    fdir, wdir, dst = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    F, W = load(fdir), load(wdir)
    kernels = {}
    for k in F:
        f, w = F[k], W.get(k, [])
        n = min(len(f), len(w)) if w else len(f)
        if n == 0:
            continue
        fetch = [2.0 * 1024.0 * x for x in f[:n]]
        write = [1024.0 * x for x in w[:n]] if w else [0.0] * n
        kernels[k] = {
            "launches": n,
            "fetch_bytes_per_launch_avg": sum(fetch) / n,
            "write_bytes_per_launch_avg": sum(write) / n,
            "hbm_bytes_per_launch_avg": (sum(fetch) + sum(write)) / n,
            "hbm_bytes_per_launch_max": max(a + b for a, b in zip(fetch, write)),
        }
    json.dump({"note": note, "correction": "fetch = 2 * FETCH_SIZE KiB (gfx950), write = WRITE_SIZE KiB",
               "kernels": kernels}, open(dst, "w"), indent=1)
    for k, v in kernels.items():
        print("%-70s %4d launches  avg %8.1f MB/launch" % (k[:70], v["launches"], v["hbm_bytes_per_launch_avg"] / 1e6))


if __name__ == "__main__":
    main()
