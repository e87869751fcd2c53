"""The GPU lane engine (bwtc_amd/csrc/gpu_lanes.hip): range-coder chains of the 'B' coder, one GPU lane per section,
byte for byte what the host's scalar loop (wavelet_rc.cpp, runChainW = BitEncoder, BitCoders.cpp:59-113) writes."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _w(rng, n, kind):
    """w-elements: bit << 15 | probability (1 .. 4095) of the coded bit."""
    if kind == "uniform":
        p = rng.integers(1, 4096, n)
    elif kind == "skewed":                       # mostly near-certain bits, as a text block's models give them
        p = np.clip((4095 * rng.beta(8, 0.6, n)).astype(np.int64), 1, 4095)
    else:                                        # always the extreme probabilities
        p = rng.choice([1, 2, 4094, 4095], n)
    bit = rng.integers(0, 2, n)
    return (bit << 15 | p).astype(np.uint16)


@pytest.mark.parametrize("kind", ["uniform", "skewed", "extreme"])
def test_gpu_lanes_give_the_hosts_bytes(hip_ctx, kind):
    rng = np.random.default_rng(len(kind))
    n = 6_000_000
    w = _w(rng, n, kind)
    # chains of every length: below the engine's floor (coded on the host), a few elements, unaligned starts and ends,
    # more chains than lanes, one chain of two million elements (several slices)
    cuts = sorted(set([0, 1, 9, 70_000, 140_007, 2_140_013, 2_140_014] + [int(x) for x in rng.integers(2_200_000, n, 400)] + [n]))
    want = hip_ctx.test_gpu_lanes(w, cuts, mode=1)
    got = hip_ctx.test_gpu_lanes(w, cuts, mode=0)
    assert len(got) == len(want) == len(cuts) - 1
    for j, (a, b) in enumerate(zip(got, want)):
        assert a == b, (kind, j, cuts[j], cuts[j + 1], len(a), len(b))


def test_gpu_lanes_rate(hip_ctx):
    """Not a pass/fail figure: what a lane makes of a long chain (printed)."""
    rng = np.random.default_rng(7)
    n = 200_000_000
    w = _w(rng, n, "skewed")
    cuts = [int(x) for x in np.linspace(0, n, 201)]
    t0 = time.perf_counter()
    got = hip_ctx.test_gpu_lanes(w, cuts, mode=0)
    t1 = time.perf_counter()
    want = hip_ctx.test_gpu_lanes(w, cuts, mode=1)
    t2 = time.perf_counter()
    assert got == want
    per = n // 200
    print("200 chains of %d elements: GPU lanes %.3f s (%.1f ns per element and lane, upload included), host scalar %.3f s (%.2f ns per element)"
          % (per, t1 - t0, 1e9 * (t1 - t0) / per, t2 - t1, 1e9 * (t2 - t1) / n))
