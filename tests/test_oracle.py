"""CPU tests: the oracle (oracle/bwtc_oracle.c) against the golden vectors generated from
the reference's libdivsufsort, the reference's own known-answer asserts, the golden stream
recorded from the reference's Compressor, and (when present) the reference build itself."""
import base64
import hashlib
import json
import os

import numpy as np
import pytest

from bwtc_amd import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def test_oracle_matches_small_goldens(oracle):
    cases = _load("bwt_small.json")["cases"]
    assert len(cases) >= 50
    for c in cases:
        data = np.frombuffer(base64.b64decode(c["input"]), np.uint8)
        bwt, lf, freqs = oracle.oracle_bwt_block(data, c["sp"])
        assert bwt.tobytes() == base64.b64decode(c["bwt"]), c["name"]
        assert [int(x) for x in lf] == c["lf"], c["name"]
        want = np.zeros(256, np.uint32)
        for k, v in c["freqs_nonzero"].items():
            want[int(k)] = v
        assert (freqs == want).all(), c["name"]
        rc, inv = oracle.oracle_inverse_bwt_block(bwt, lf)
        assert rc == 0 and inv.tobytes() == data.tobytes(), c["name"]


def test_oracle_matches_large_golden_1mib(oracle):
    c = [x for x in _load("bwt_large.json")["cases"] if x["name"] == "C1_random_1MiB"][0]
    d = getattr(synth, c["gen"])(c["size"], c["seed"])
    assert hashlib.sha256(d.tobytes()).hexdigest() == c["input_sha256"]
    bwt, lf, _ = oracle.oracle_bwt_block(d, c["sp"])
    assert hashlib.sha256(bwt.tobytes()).hexdigest() == c["bwt_sha256"]
    assert [int(x) for x in lf] == c["lf"]


def test_generators_are_stable():
    for c in _load("bwt_large.json")["cases"]:
        if c["size"] > (4 << 20):
            continue
        d = getattr(synth, c["gen"])(c["size"], c["seed"])
        assert hashlib.sha256(d.tobytes()).hexdigest() == c["input_sha256"], c["name"]


def test_oracle_vs_reference_build_random(oracle):
    if oracle.ref() is None:
        pytest.skip("oracle/_ref not built (reference sources absent)")
    rng = np.random.default_rng(7)
    for it in range(400):
        n = int(rng.integers(1, 1200))
        sigma = int(rng.choice([1, 2, 3, 4, 16, 256]))
        d = rng.integers(0, sigma, n).astype(np.uint8)
        if it % 3 == 0:
            d = np.tile(d[:max(1, n // 7)], 8)[:n]
        sp = int(rng.choice([1, 2, 3, 8, 256, 300]))
        a = oracle.oracle_bwt_block(d, sp)
        b = oracle.ref_bwt_block(d, sp)
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all()


def test_suffix_array_is_sorted(oracle):
    # the property test/SaisTest.cpp:55-70 checks: SA sorted, SA[0] == size for 0-terminated input
    rng = np.random.default_rng(3)
    for _ in range(10):
        n = int(rng.integers(2, 400))
        t = np.concatenate([rng.integers(0, 4, n).astype(np.uint8), [0]]).astype(np.uint8)
        sa = oracle.oracle_suffix_array(t)
        assert sa[0] == n
        b = t.tobytes()
        suf = [b[i:] for i in sa]
        assert suf == sorted(suf)


def test_reference_known_answers(oracle):
    import ctypes
    ka = _load("utils_known_answers.json")
    L = oracle.lib()
    for c in ka["run_frequencies"]:
        src = np.frombuffer(c["input"].encode(), np.uint8)
        rf = np.zeros(256, np.uint64)
        L.orc_run_frequencies(oracle._ptr(rf), oracle._ptr(src), src.size)
        want = np.zeros(256, np.uint64)
        for k, v in c["expect"].items():
            want[ord(k)] = v
        assert (rf == want).all(), c["ref"]
    for v in ka["packed_integers"]["values"]:
        nb = ctypes.c_int(0)
        p = L.orc_pack_integer(v, ctypes.byref(nb))
        raw = [(p >> (8 * i)) & 0xFF for i in range(nb.value)]
        back = 0
        for i, b in enumerate(raw):
            back |= (b & 0x7F) << (7 * i)
            assert bool(b & 0x80) == (i + 1 < len(raw))
        assert back == v
    for c in ka["huffman_lengths"]:
        fr = np.zeros(256, np.uint64)
        for k, v in c["freqs"].items():
            fr[ord(k)] = v
        cl = np.zeros(256, np.uint32)
        L.orc_huffman_lengths(oracle._ptr(fr), oracle._ptr(cl))
        order = sorted(c["freqs"].items(), key=lambda kv: (kv[1], ord(kv[0])))
        assert [int(cl[ord(k)]) for k, _ in order] == c["sorted_lengths"], c["ref"]
        for k, v in c.get("by_symbol", {}).items():
            assert cl[ord(k)] == v
    bits = np.zeros(64, np.uint8)
    for n, lo, hi, want in ka["binary_code"]["cases"]:
        nb = L.orc_binary_code(n, lo, hi, oracle._ptr(bits))
        assert "".join(str(int(b)) for b in bits[:nb]) == want, (n, lo, hi)


def test_golden_stream_H(oracle):
    c = [x for x in _load("streams.json")["cases"] if x["coder"] == "H"][0]
    s = oracle.oracle_compress_H(c["input_ascii"].encode(), c["block_size"], c["sp"])
    assert s.tobytes() == bytes.fromhex(c["stream_hex"])
    assert oracle.oracle_decompress_H(s, 64).tobytes() == c["input_ascii"].encode()


def test_compress_roundtrip_like_reference_test(oracle):
    # test/CompressorAndDecompressorTest.cpp:61-205: random and repetitive data, single and
    # multiple blocks, 1..30 starting points, coder 'H'.
    rng = np.random.default_rng(11)
    for size, block, sp in [(100, 100, 1), (1000, 300, 3), (100000, 100000, 8), (100000, 18500, 30),
                            (50000, 7000, 2)]:
        for rep in (False, True):
            d = rng.integers(0, 256, size).astype(np.uint8)
            if rep:
                d = np.tile(d[:size // 50 + 1], 51)[:size]
            s = oracle.oracle_compress_H(d, block, sp)
            r = oracle.oracle_decompress_H(s, size + 8)
            assert r is not None and r.tobytes() == d.tobytes()


def test_runs_and_characters_known_answers(oracle):
    # the run counts of test/UtilsTest.cpp:79-129 hold for calculateRunsAndCharacters too
    ka = _load("utils_known_answers.json")
    for c in ka["run_frequencies"]:
        rf, runs, dist = oracle.oracle_runs_and_characters(np.frombuffer(c["input"].encode(), np.uint8))
        for k, v in c["expect"].items():
            assert rf[ord(k)] == v
        assert runs == sum(c["expect"].values()) == sum(dist.values())
        assert sum(l * n for l, n in dist.items()) == len(c["input"])
    rf, runs, dist = oracle.oracle_runs_and_characters(np.frombuffer(b"abdbcarraaa", np.uint8))
    assert runs == 8 and dist == {1: 6, 2: 1, 3: 1}
