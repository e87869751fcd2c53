"""Per-kernel means of the counters in rocprofv3 counter_collection csv files.

usage: python scripts/dev/pmc_table.py a_counter_collection.csv [b_counter_collection.csv ...] [--grep name]
"""
import csv, re, sys
from collections import defaultdict


def main():
    files = [a for a in sys.argv[1:] if not a.startswith('--')]
    pat = None
    if '--grep' in sys.argv:
        pat = sys.argv[sys.argv.index('--grep') + 1]
        files = [f for f in files if f != pat]
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        per = defaultdict(dict)
        for r in csv.DictReader(open(f)):
            nm = re.sub(r'\(.*', '', r['Kernel_Name']).replace('bwtc_hip::', '').replace('void ', '')
            per[(r['Dispatch_Id'], nm)][r['Counter_Name']] = float(r['Counter_Value'])
        for (d, nm), cs in per.items():
            for c, v in cs.items():
                acc[nm][c].append(v)
    names = sorted({c for k in acc for c in acc[k]})
    for nm in sorted(acc):
        if pat and not re.search(pat, nm):
            continue
        print(nm)
        for c in names:
            if c in acc[nm]:
                v = acc[nm][c]
                print(f"    {c:24s} {sum(v) / len(v):16.0f}   (n={len(v)})")


if __name__ == '__main__':
    main()
