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


def _synth():
    from bwtc_amd import synth
    return synth


def structured(rng, n):
    """Inputs that stress the suffix sorter: long repeats, periods, tiny alphabets, Fibonacci words."""
    kind = int(rng.integers(0, 7))
    if kind == 0:
        d = np.full(n, int(rng.integers(0, 256)))
    elif kind == 1:
        p = rng.integers(0, 256, int(rng.integers(1, 40)))
        d = np.tile(p, n // p.size + 1)[:n].copy()
        d[rng.random(n) < float(rng.uniform(0, 0.002))] = int(rng.integers(0, 256))
    elif kind == 2:
        a, b = [0], [0, 1]
        while len(b) < n:
            a, b = b, b + a
        d = np.array(b[:n]) + int(rng.integers(0, 200))
    elif kind == 3:
        d = rng.integers(0, int(rng.integers(2, 5)), n)
    elif kind == 4:
        base = rng.integers(0, 256, max(1, n // int(rng.integers(2, 50))))
        d = np.concatenate([base] * (n // base.size + 1))[:n].copy()
        k = int(rng.integers(0, 20))
        if k:
            d[rng.integers(0, n, k)] = rng.integers(0, 256, k)
    elif kind == 5:
        d = _synth().gen_text(n, int(rng.integers(1, 1 << 30)))
    else:
        d = _synth().gen_dna(n, int(rng.integers(1, 1 << 30)))
    return kind, np.ascontiguousarray(d, dtype=np.uint8)
