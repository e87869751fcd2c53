"""Small byte blocks of varied shape for sweeps over the entropy coders: alphabet size, run
structure, skew, long runs, sparse symbols, many sections."""
import numpy as np


def varied_blocks(count, max_size, seed):
    rng = np.random.default_rng(seed)
    for case in range(count):
        n = int(rng.integers(1, max_size))
        kind = case % 8
        if kind == 0:
            d = rng.integers(0, int(rng.integers(1, 257)), n)
        elif kind == 1:                                   # runs with geometric lengths
            k = max(1, n // int(rng.integers(2, 200)))
            d = np.repeat(rng.integers(0, int(rng.integers(1, 40)), k),
                          rng.geometric(1.0 / int(rng.integers(2, 400)), k))[:n]
        elif kind == 2:                                   # heavy skew
            d = np.minimum(rng.geometric(float(rng.uniform(0.05, 0.9)), n) - 1, 255)
        elif kind == 3:                                   # two symbols far apart + rare others
            d = np.where(rng.random(n) < 0.97, rng.choice([3, 250], n), rng.integers(0, 256, n))
        elif kind == 4:                                   # one very long run inside noise
            d = rng.integers(0, 6, n)
            a = int(rng.integers(0, n))
            d[a:a + int(rng.integers(1, n + 1))] = 2
        elif kind == 5:                                   # alternating pattern with defects
            d = np.arange(n) % int(rng.integers(2, 7))
            d[rng.random(n) < 0.01] = 9
        elif kind == 6:                                   # all symbols present, uniform: many sections
            d = rng.permutation(np.arange(n) % 256)
        else:                                             # power-law run lengths
            k = max(1, n // 50)
            d = np.repeat(rng.integers(0, 12, k),
                          np.minimum((rng.pareto(0.8, k) + 1).astype(np.int64), 5000))[:n]
        d = np.ascontiguousarray(d, dtype=np.uint8)
        if d.size:
            yield case, kind, d
