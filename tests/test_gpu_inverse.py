"""GPU tests of the inverse transform (list ranking over the LF cycle) through the C ABI:
against the oracle's LF walk, as a round trip behind the GPU forward transform, and in the
procedure of the reference's test/InverseBwtTest.cpp:51-114."""
import base64
import hashlib
import json
import os

import numpy as np
import pytest

from bwtc_amd import hip, synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_inverse_of_reference_goldens(hip_ctx):
    for c in json.load(open(os.path.join(G, "bwt_small.json")))["cases"]:
        data = base64.b64decode(c["input"])
        bwt = np.frombuffer(base64.b64decode(c["bwt"]), np.uint8)
        back = hip_ctx.inverse_bwt_block(bwt, c["lf"])
        assert back.tobytes() == data, c["name"]


def test_forward_inverse_roundtrip_random(hip_ctx, oracle):
    rng = np.random.default_rng(4242)
    for it in range(60):
        n = int(rng.integers(1, 300000))
        sigma = int(rng.choice([1, 2, 4, 256]))
        d = rng.integers(0, sigma, n).astype(np.uint8)
        if it % 3 == 0:
            d = np.tile(d[:max(1, n // 13)], 14)[:n]
        sp = int(rng.integers(1, 301))
        bwt, lf, _ = hip_ctx.bwt_block(d, sp)
        back = hip_ctx.inverse_bwt_block(bwt, lf)
        assert (back == d).all(), (it, n, sigma, sp)
        rc, oinv = oracle.oracle_inverse_bwt_block(bwt, lf)
        assert rc == 0 and (oinv == back).all()


def test_inverse_rejects_bad_lf_power(hip_ctx):
    d = synth.gen_text(100000, 3)
    bwt, lf, _ = hip_ctx.bwt_block(d, 8)
    bad = lf.copy()
    bad[3] = (bad[3] + 1) % d.size
    with pytest.raises(hip.BwtcHipError):
        hip_ctx.inverse_bwt_block(bwt, bad)


def test_inverse_large_text(hip_ctx):
    size = 64 << 20
    d = synth.gen_text(size, 3)
    bwt, lf, _ = hip_ctx.bwt_block(d, 8)
    back = hip_ctx.inverse_bwt_block(bwt, lf)
    st = hip_ctx.stats()
    assert hashlib.sha256(back.tobytes()).digest() == hashlib.sha256(d.tobytes()).digest()
    print("inverse 64 MiB text: %.2f ms device -> %.0f MB/s" % (st.ms_total, size / 1e6 / (st.ms_total / 1e3)))
