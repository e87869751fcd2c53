import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bwtc_amd import hip
ctx = hip.Context(0, 64 << 20)
for n, hi in [(30000001, 1000), (30000001, 2), (5000000, 1000), (1 << 20, 1000), (300000, 1000)]:
    rng = np.random.default_rng(n)
    d = rng.integers(0, hi, n).astype(np.uint32)
    got = ctx.test_scan(d.copy())
    want = np.concatenate([[0], np.cumsum(d[:-1], dtype=np.uint64)]).astype(np.uint32)
    bad = np.flatnonzero(got != want)
    print(n, hi, "mismatches", bad.size, "first", bad[:5], "tiles", bad[:5] // 4096 if bad.size else "", (got[bad[:3]].astype(np.int64) - want[bad[:3]].astype(np.int64)) if bad.size else "")
