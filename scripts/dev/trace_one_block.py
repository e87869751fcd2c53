"""Prints the kernel sequence of the last block of a rocprofv3 kernel trace (csv), with durations.

usage: python scripts/dev/trace_one_block.py <..._kernel_trace.csv> [min_us]
"""
import csv, re, sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'k_load_hist' in r['Kernel_Name']]
    a = idx[-1]
    t0 = int(rows[a]['Start_Timestamp'])
    total = 0.0
    for r in rows[a:]:
        nm = re.sub(r'\(.*', '', r['Kernel_Name']).replace('bwtc_hip::', '').replace('void ', '')
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        total += d
        if d >= min_us:
            print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:8.2f} ms {d:8.1f} us  {nm}")
    print(f"kernels of the block: {total / 1e3:.2f} ms")


if __name__ == '__main__':
    main()
