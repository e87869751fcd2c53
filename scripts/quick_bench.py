#!/usr/bin/env python3
"""Ad-hoc timing of the block transform through the host-pointer ABI (device time from the
context's HIP events).  Usage: quick_bench.py [size_MiB ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bwtc_amd import hip, synth  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
kinds = [a[2:] for a in sys.argv[1:] if a.startswith("--")] or ["text", "dna", "random"]
sizes = [int(a) for a in args] or [16, 64]
ctx = hip.Context(0, (max(sizes) << 20) + 64)
def gen_zeros(n, seed):
    return np.zeros(n, np.uint8)


def gen_period(n, seed):
    return np.tile(np.frombuffer(b"abcabcabd" * 113, np.uint8), n // 1017 + 1)[:n].copy()


def gen_reptext(n, seed):
    base = synth.gen_text(1 << 20, seed)
    return np.tile(base, n // base.size + 1)[:n].copy()


for kind, gen, seed in [("text", synth.gen_text, 3), ("dna", synth.gen_dna, 2), ("random", synth.gen_random_bytes, 1),
                        ("zeros", gen_zeros, 0), ("period", gen_period, 0), ("reptext", gen_reptext, 3)]:
    if kind not in kinds:
        continue
    for mib in sizes:
        d = gen(mib << 20, seed)
        best = None
        for rep in range(3):
            t0 = time.time()
            bwt, lf, fr = ctx.bwt_block(d, 8)
            wall = time.time() - t0
            st = ctx.stats()
            if best is None or st.ms_total < best[0]:
                best = (st.ms_total, st.ms_sort, st.rounds, st.active_sum / st.n, wall, st.sort_pass_items / st.n)
        inv = ctx.inverse_bwt_block(bwt, lf)
        assert (inv == d).all()
        ims = ctx.stats().ms_total
        print("%-6s %4d MiB: inverse %7.2f ms -> %8.1f MB/s" % (kind, mib, ims, (mib << 20) / 1e6 / (ims / 1e3)), flush=True)
        print("%-6s %4d MiB: device %8.2f ms (sort %8.2f) rounds %2d R_eff %.2f passes/N %.1f  -> %8.1f MB/s (wall %.2fs)"
              % (kind, mib, best[0], best[1], best[2], best[3], best[5], (mib << 20) / 1e6 / (best[0] / 1e3), best[4]), flush=True)
