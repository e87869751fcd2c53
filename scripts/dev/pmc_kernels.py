"""One line per kernel from the two SQ counter passes of scripts/dev/pmc_sq.sh: waves, instructions
per wave, share of the busy cycles the VALU was issuing, wave lifetime, share of it spent waiting.

usage: python scripts/dev/pmc_kernels.py gpurun_out/pmc_sq [min_grid]
"""
import csv, re, sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    min_grid = int(sys.argv[2]) if len(sys.argv) > 2 else 20000000
    acc = defaultdict(lambda: defaultdict(list))
    for f in (root + '/a/a_counter_collection.csv', root + '/b/b_counter_collection.csv'):
        per = defaultdict(dict)
        for r in csv.DictReader(open(f)):
            nm = re.sub(r'\(.*', '', r['Kernel_Name']).replace('bwtc_hip::', '').replace('void ', '')
            per[(r['Dispatch_Id'], nm, r['Grid_Size'])][r['Counter_Name']] = float(r['Counter_Value'])
        for (d, nm, g), cs in per.items():
            if int(g) < min_grid:
                continue
            for c, v in cs.items():
                acc[nm][c].append(v)
    print(f"{'kernel':58s} {'waves':>8s} {'valu/w':>7s} {'lds/w':>6s} {'vmem/w':>6s} {'VALUbusy':>8s} {'waveq/w':>8s} {'wait%':>6s} {'busyMcyc':>8s}")
    for nm, c in sorted(acc.items()):
        m = lambda k: sum(c[k]) / len(c[k]) if k in c else 0
        w = m('SQ_WAVES')
        if not w:
            continue
        busy = m('SQ_BUSY_CYCLES') / 32
        print(f"{nm[:58]:58s} {w:8.0f} {m('SQ_INSTS_VALU') / w:7.0f} {m('SQ_INSTS_LDS') / w:6.0f} "
              f"{(m('SQ_INSTS_VMEM_RD') + m('SQ_INSTS_VMEM_WR')) / w:6.0f} "
              f"{m('SQ_ACTIVE_INST_VALU') / 1024 / (busy / 4) if busy else 0:8.2f} {m('SQ_WAVE_CYCLES') / w:8.0f} "
              f"{m('SQ_WAIT_ANY') / m('SQ_WAVE_CYCLES') * 100 if m('SQ_WAVE_CYCLES') else 0:6.1f} {busy / 1e6:8.2f}")


if __name__ == '__main__':
    main()
