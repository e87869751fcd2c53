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


def test_inverse_fails_cleanly_on_corrupt_headers_and_data(hip_ctx):
    """What a damaged file can hand the inverse transform: LF powers outside the block, a wrong
    end-of-block row (LF then splits into several cycles, some without a splitter row), damaged
    bytes.  Every case must come back with an error code (-4 / -1) or a wrong-but-finite result --
    never a fault or an endless walk; one run each, bounded by the suite's timeout."""
    d = synth.gen_text(300000, 11)
    bwt, lf, _ = hip_ctx.bwt_block(d, 8)
    for k in (1, 7):
        bad = lf.copy()
        bad[k] = d.size + 1 + k                       # not a row of this block
        with pytest.raises(hip.BwtcHipError) as err:
            hip_ctx.inverse_bwt_block(bwt, bad)
        assert "-4" in str(err.value)
    bad = lf.copy()
    bad[0] = (int(lf[0]) + 12345) % d.size            # wrong end-of-block row, n_lf > 1
    with pytest.raises(hip.BwtcHipError) as err:
        hip_ctx.inverse_bwt_block(bwt, bad)
    assert "-4" in str(err.value)
    # wrong end-of-block row with a single LF power: nothing to check it against except the cycle
    # structure (rows that are not on the walk from row 0)
    bad1 = np.array([(int(lf[0]) + 777) % d.size], np.uint32)
    try:
        back = hip_ctx.inverse_bwt_block(bwt, bad1)
        assert back.size == d.size                    # accepted: LF happened to stay one cycle
    except hip.BwtcHipError as e:
        assert "-4" in str(e)
    # damaged bytes: a few symbols changed
    hurt = bwt.copy()
    hurt[1000:1010] ^= 0x55
    try:
        back = hip_ctx.inverse_bwt_block(hurt, lf)
        assert not (back == d).all()
    except hip.BwtcHipError as e:
        assert "-4" in str(e)
    # and the context still works afterwards
    assert (hip_ctx.inverse_bwt_block(bwt, lf) == d).all()


def test_inverse_large_text(hip_ctx):
    size = 64 << 20
    d = synth.gen_text(size, 3)
    bwt, lf, _ = hip_ctx.bwt_block(d, 8)
    back = hip_ctx.inverse_bwt_block(bwt, lf)
    st = hip_ctx.stats()
    assert hashlib.sha256(back.tobytes()).digest() == hashlib.sha256(d.tobytes()).digest()
    print("inverse 64 MiB text: %.2f ms device -> %.0f MB/s" % (st.ms_total, size / 1e6 / (st.ms_total / 1e3)))
