"""GPU parity tests (run on the MI355X box with -m gpu).  Everything goes through the C ABI
(bwtc_amd/hip.py -> libbwtc_hip.so); the oracle and the golden fixtures are the checkers."""
import base64
import hashlib
import json
import os

import numpy as np
import pytest

from bwtc_amd import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


# ---- primitives --------------------------------------------------------------------------

@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 4095, 4096, 4097, 100000, (1 << 20) + 3, 30000001])
def test_scan(hip_ctx, n):
    rng = np.random.default_rng(n)
    d = rng.integers(0, 1000 if n < 4000000 else 100, n).astype(np.uint32)      # the total fits 32 bits (the scan's contract)
    got = hip_ctx.test_scan(d)
    want = np.concatenate([np.zeros(1, np.uint64), np.cumsum(d[:-1], dtype=np.uint64)]).astype(np.uint32)
    assert (got == want).all()


@pytest.mark.parametrize("n", [1, 2, 100, 8191, 8192, 8193, 300001])
@pytest.mark.parametrize("mode", ["random", "few", "equal"])
def test_sort_u32(hip_ctx, n, mode):
    rng = np.random.default_rng(n)
    if mode == "random":
        k = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    elif mode == "few":
        k = (rng.integers(0, 3, n).astype(np.uint32) * np.uint32(0x01010101))
    else:
        k = np.full(n, 0xDEADBEEF, np.uint32)
    v = np.arange(n, dtype=np.uint32)
    gk, gv = hip_ctx.test_sort(k, v, 32)
    order = np.argsort(k, kind="stable")
    assert (gk == k[order]).all()
    assert (gv == v[order]).all()          # stability


@pytest.mark.parametrize("n,nbits", [(1, 64), (5, 64), (4096, 64), (4097, 40), (250001, 57), (250001, 13)])
def test_sort_u64(hip_ctx, n, nbits):
    rng = np.random.default_rng(n + nbits)
    k = rng.integers(0, 1 << 63, n, dtype=np.uint64)
    if nbits < 64:
        k &= np.uint64((1 << nbits) - 1)
    k[::7] = k[0]
    v = np.arange(n, dtype=np.uint32)
    gk, gv = hip_ctx.test_sort(k, v, nbits)
    order = np.argsort(k, kind="stable")
    assert (gk == k[order]).all()
    assert (gv == v[order]).all()


# ---- suffix array (the property test/SaisTest.cpp:55-70 checks) ----------------------------

def test_suffix_array_small(hip_ctx, oracle):
    rng = np.random.default_rng(5)
    for it in range(40):
        n = int(rng.integers(1, 3000))
        sigma = int(rng.choice([1, 2, 4, 256]))
        t = np.concatenate([rng.integers(0, sigma, n).astype(np.uint8), [0]]).astype(np.uint8)
        sa = hip_ctx.suffix_array(t)
        assert sa[0] == n
        assert (sa == oracle.oracle_suffix_array(t)).all(), (it, n, sigma)


# ---- block transform against the reference-generated goldens ------------------------------

def test_bwt_block_small_goldens(hip_ctx):
    for c in _load("bwt_small.json")["cases"]:
        data = np.frombuffer(base64.b64decode(c["input"]), np.uint8)
        bwt, lf, freqs = hip_ctx.bwt_block(data, c["sp"])
        assert bwt.tobytes() == base64.b64decode(c["bwt"]), c["name"]
        assert [int(x) for x in lf] == c["lf"], c["name"]
        want = np.zeros(256, np.uint32)
        for k, v in c["freqs_nonzero"].items():
            want[int(k)] = v
        assert (freqs == want).all(), c["name"]


def test_bwt_block_large_goldens(hip_ctx):
    for c in _load("bwt_large.json")["cases"]:
        if c["size"] > (64 << 20):
            continue                      # C3_text_256MiB: test_baseline_full_size_256MiB_text* below
        d = getattr(synth, c["gen"])(c["size"], c["seed"])
        assert hashlib.sha256(d.tobytes()).hexdigest() == c["input_sha256"]
        bwt, lf, freqs = hip_ctx.bwt_block(d, c["sp"])
        assert hashlib.sha256(bwt.tobytes()).hexdigest() == c["bwt_sha256"], c["name"]
        assert [int(x) for x in lf] == c["lf"], c["name"]
        assert (freqs == np.bincount(d, minlength=256)).all()


def test_bwt_block_random_vs_oracle(hip_ctx, oracle):
    # test/InverseBwtTest.cpp:51-114 and test/LFpowersTest.cpp:49-181 in spirit: random
    # blocks, every alphabet size incl. 0 bytes, random starting points; plus the inverse.
    rng = np.random.default_rng(99)
    for it in range(120):
        n = int(rng.integers(1, 20000))
        sigma = int(rng.choice([1, 2, 3, 26, 256]))
        d = rng.integers(0, sigma, n).astype(np.uint8)
        if it % 4 == 0:
            d = np.tile(d[:max(1, n // 11)], 12)[:n]
        sp = int(rng.integers(1, 301))
        a = hip_ctx.bwt_block(d, sp)
        b = oracle.oracle_bwt_block(d, sp)
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all(), (it, n, sigma, sp)
        rc, inv = oracle.oracle_inverse_bwt_block(a[0], a[1])
        assert rc == 0 and (inv == d).all()


def test_bwt_raw_leaves_pidx_slot_untouched(hip_ctx, oracle):
    # raw virtual: BWTransform::doTransform(begin, length, LF, freqs), divsufsort.c:506-512
    rng = np.random.default_rng(17)
    for n in [2, 3, 10, 1000, 70000]:
        t = np.concatenate([rng.integers(0, 5, n - 1).astype(np.uint8), [0]]).astype(np.uint8)
        n_lf = 5 if n > 300 else 1
        got, lf, fr = hip_ctx.bwt_raw(t, n_lf)
        want = t.copy()
        wlf = np.zeros(n_lf, np.uint32)
        wfr = np.zeros(256, np.uint32)
        oracle.lib().orc_bwt_raw(oracle._ptr(want), n, oracle._ptr(wlf), n_lf, oracle._ptr(wfr))
        assert (got == want).all() and (lf == wlf).all() and (fr == wfr).all()
        assert got[lf[0]] == t[lf[0]]


def test_degenerate_blocks(hip_ctx, oracle):
    for d in [np.zeros(1, np.uint8), np.zeros(70000, np.uint8), np.full(50000, 255, np.uint8),
              np.tile(np.array([1, 0], np.uint8), 40000),
              np.tile(np.frombuffer(b"abcabcabd", np.uint8), 9000)]:
        a = hip_ctx.bwt_block(d, 8)
        b = oracle.oracle_bwt_block(d, 8)
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all()


def test_gram_coded_initial_keys_vs_oracle(oracle):
    """Blocks of 4 Mi suffixes and more make their initial sort key from dense gram codes when that
    saves radix passes (bwt_engine.hip, K2g).  Alphabets and shapes that steer the plan differently
    -- text, text with zero bytes (the terminator then has a code of its own), a sparse two-symbol
    block (22-character grams), a skewed 200-symbol alphabet (2-character grams), a noisy period,
    a block ending in a run of its smallest symbol -- all against the reference's sorter."""
    from bwtc_amd import hip
    size = 5 << 20
    rng = np.random.default_rng(2024)
    text = synth.gen_text(size, 11)
    with_zeros = text.copy()
    with_zeros[with_zeros == 32] = 0
    sparse = (rng.random(size) < 0.03).astype(np.uint8) * 7
    zipf = np.minimum(rng.zipf(1.3, size), 200).astype(np.uint8)
    period = np.tile(np.frombuffer(b"abracad", np.uint8), size // 7 + 1)[:size].copy()
    period[rng.integers(0, size, 200)] = 122
    tail = text.copy()
    tail[-100:] = text.min()
    blocks = [("text", text), ("with_zeros", with_zeros), ("sparse", sparse), ("zipf", zipf),
              ("period", period), ("tail", tail)]
    with hip.Context(0, size) as ctx:
        for name, d in blocks:
            a = ctx.bwt_block(d, 5)
            b = oracle.oracle_bwt_block(d, 5)
            assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all(), name


def test_full_size_text_roundtrip(hip_ctx, oracle):
    # 64 MiB text block: too big for the oracle's sorter, so check the size-independent
    # properties: the inverse transform reproduces the input (and every LF power lies on the
    # LF walk), and freqs is the byte histogram.
    size = 64 << 20
    d = synth.gen_text(size, 3)
    bwt, lf, freqs = hip_ctx.bwt_block(d, 8)
    assert (freqs == np.bincount(d, minlength=256)).all()
    rc, inv = oracle.oracle_inverse_bwt_block(bwt, lf)
    assert rc == 0
    assert hashlib.sha256(inv.tobytes()).digest() == hashlib.sha256(d.tobytes()).digest()
    st = hip_ctx.stats()
    print("64MiB text: %.1f ms total, %.1f ms sort, rounds %d, R_eff %.2f" %
          (st.ms_total, st.ms_sort, st.rounds, st.active_sum / st.n))


def test_worst_case_blocks_at_64MiB(oracle):
    """The inputs prefix doubling likes least, at a size where every round takes the window-ordered
    route: an all-zero block (one group until the last round, log2 N rounds), a period-9 block and
    a 1 MiB text repeated 64 times (21-23 rounds).  Checked by the GPU inverse (which verifies every
    LF power against its own ranking) and by the oracle's inverse on the all-zero block."""
    from bwtc_amd import hip
    size = 64 << 20
    blocks = [("zeros", np.zeros(size, np.uint8)),
              ("period9", np.tile(np.frombuffer(b"abcabcabd", np.uint8), size // 9 + 1)[:size].copy()),
              ("reptext", np.tile(synth.gen_text(1 << 20, 3), 64))]
    with hip.Context(0, size) as ctx:
        for name, d in blocks:
            bwt, lf, freqs = ctx.bwt_block(d, 8)
            st = ctx.stats()
            assert (freqs == np.bincount(d, minlength=256)).all(), name
            back = ctx.inverse_bwt_block(bwt, lf)
            assert back.size == size and (back == d).all(), name
            if name == "zeros":
                assert (bwt == 0).all() and int(lf[0]) == size      # the end-of-block row is the last one
                rc, inv = oracle.oracle_inverse_bwt_block(bwt, lf)
                assert rc == 0 and (inv == d).all()
            print("%s 64MiB: %.1f ms device, %d rounds" % (name, st.ms_total, st.rounds))


def test_baseline_full_size_256MiB_text(oracle):
    """BASELINE.json config 3 at full size (256 MiB text block, 8 starting points): the
    size-independent properties -- inverse transform reproduces the input with every LF power
    on the LF walk, freqs is the byte histogram -- plus, when the reference build is present,
    bit-equality with the reference's divbwtf."""
    from bwtc_amd import hip
    size = 256 << 20
    d = synth.gen_text(size, 3)
    with hip.Context(0, size) as ctx:
        bwt, lf, freqs = ctx.bwt_block(d, 8)
        st = ctx.stats()
    assert lf.size == 8
    assert (freqs == np.bincount(d, minlength=256)).all()
    pin = [c for c in _load("bwt_large.json")["cases"] if c["name"] == "C3_text_256MiB"][0]
    assert hashlib.sha256(bwt.tobytes()).hexdigest() == pin["bwt_sha256"] and [int(x) for x in lf] == pin["lf"]
    rc, inv = oracle.oracle_inverse_bwt_block(bwt, lf)
    assert rc == 0
    assert hashlib.sha256(inv.tobytes()).digest() == hashlib.sha256(d.tobytes()).digest()
    del inv
    if oracle.ref() is not None:
        rb, rlf, rfr = oracle.ref_bwt_block(d, 8)
        assert hashlib.sha256(rb.tobytes()).digest() == hashlib.sha256(bwt.tobytes()).digest()
        assert (rlf == lf).all() and (rfr == freqs).all()
    print("256MiB text: %.1f ms device, rounds %d, R_eff %.2f" % (st.ms_total, st.rounds, st.active_sum / st.n))


def test_baseline_full_size_256MiB_text_B_record_equals_oracle(oracle):
    """BASELINE.json config 3 through the default coder at FULL size: the 'B' record of the
    256 MiB text block (544 M steps through the stream kernels, the huge-group scalar tasks at
    their real threshold) byte for byte against the oracle's literal WaveletEncoder, and against
    the SHA-256 committed in tests/golden/bwt_large.json.  The oracle encodes the transformed
    block in about half a minute on one core."""
    import json
    from bwtc_amd import hip
    size = 256 << 20
    d = synth.gen_text(size, 3)
    with hip.Context(0, size) as ctx:
        ctx.wavelet_reset()
        rec, bwt = ctx.transform_and_encode_wavelet(d, 8)
        lf, freqs = None, np.bincount(d, minlength=256).astype(np.uint32)
        bwt2, lf, freqs2 = ctx.bwt_block(d, 8)
    assert (bwt == bwt2).all() and (freqs == freqs2).all()
    del bwt2, d
    want = oracle.oracle_wavelet_encode_block(bwt, lf, freqs)
    assert rec.size == want.size, (rec.size, want.size)
    assert rec.tobytes() == want.tobytes()
    sha = hashlib.sha256(rec.tobytes()).hexdigest()
    print("256MiB text 'B' record: %d bytes, sha256 %s" % (rec.size, sha))
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bwt_large.json")))
    pins = [c for c in g["cases"] if c["name"] == "C3_text_256MiB"]
    assert pins, "C3_text_256MiB missing from bwt_large.json"
    assert pins[0]["b_record_bytes"] == rec.size and pins[0]["b_record_sha256"] == sha


def test_baseline_1GiB_single_block_round_trip():
    """BASELINE.json config 5 at full size: one 1 GiB block (N = 2^30 + 1, the largest the 31-bit
    header fields carry), forward transform and inverse transform on the GPU; the inverse also
    verifies every LF power against its own ranking.  Size-independent property, no oracle."""
    from bwtc_amd import hip
    size = 1 << 30
    d = synth.gen_dna(size, 5)
    with hip.Context(0, size) as ctx:
        bwt, lf, freqs = ctx.bwt_block(d, 8)
        assert (freqs == np.bincount(d, minlength=256)).all()
        assert lf.size == 8 and int(lf.max()) <= size
        back = ctx.inverse_bwt_block(bwt, lf)
    assert back.size == size and (back == d).all()


def test_baseline_1GiB_single_text_block_round_trip():
    """Config 5 with the C3 generator's text (seed 5): the long-key route with 16-byte items (no split index above
    512 MiB), the finisher over 37 % of the block, text rounds for its oversize groups -- forward and inverse on
    the GPU, the inverse verifying every LF power (which this route notes as suffixes become final)."""
    from bwtc_amd import hip
    size = 1 << 30
    with hip.Context(0, size) as ctx:
        d = hip.synth_into("t", 5, np.empty(size, np.uint8))
        bwt, lf, freqs = ctx.bwt_block(d, 8)
        assert ctx.stats().route & 3 == 3
        assert (freqs == np.bincount(d, minlength=256)).all()
        back = ctx.inverse_bwt_block(bwt, lf)
    assert back.size == size and (back == d).all()


@pytest.mark.skipif(os.environ.get("BWTC_TEST_1GIB") != "1", reason="set BWTC_TEST_1GIB=1 (takes minutes)")
def test_baseline_1GiB_single_block(oracle):
    """BASELINE.json config 5: one 1 GiB block, 32-bit indices, N = 2^30 + 1."""
    from bwtc_amd import hip
    size = 1 << 30
    d = synth.gen_dna(size, 5)
    with hip.Context(0, size) as ctx:
        bwt, lf, freqs = ctx.bwt_block(d, 8)
    assert (freqs == np.bincount(d, minlength=256)).all()
    rc, inv = oracle.oracle_inverse_bwt_block(bwt, lf)
    assert rc == 0
    assert hashlib.sha256(inv.tobytes()).digest() == hashlib.sha256(d.tobytes()).digest()


def test_chained_sort_variant(oracle, monkeypatch):
    """The opt-in chained single-read radix passes (BWTC_HIP_SORT=sweep; decoupled look-back
    with bounded spins) must give the same results as the default passes."""
    from bwtc_amd import hip
    monkeypatch.setenv("BWTC_HIP_SORT", "sweep")
    with hip.Context(0, 4 << 20) as ctx:
        rng = np.random.default_rng(77)
        for n, nbits in [(8193, 32), (300001, 32), (1 << 20, 20)]:
            k = rng.integers(0, 1 << nbits, n, dtype=np.uint64).astype(np.uint32)
            v = np.arange(n, dtype=np.uint32)
            gk, gv = ctx.test_sort(k, v, nbits)
            order = np.argsort(k, kind="stable")
            assert (gk == k[order]).all() and (gv == v[order]).all()
        k = rng.integers(0, 1 << 50, 700001, dtype=np.uint64)
        v = np.arange(k.size, dtype=np.uint32)
        gk, gv = ctx.test_sort(k, v, 50)
        order = np.argsort(k, kind="stable")
        assert (gk == k[order]).all() and (gv == v[order]).all()
        d = synth.gen_text(3 << 20, 3)
        a = ctx.bwt_block(d, 8)
        b = oracle.oracle_bwt_block(d[:1 << 20], 8)
        a1 = ctx.bwt_block(d[:1 << 20], 8)
        assert (a1[0] == b[0]).all() and (a1[1] == b[1]).all()
        rc, inv = oracle.oracle_inverse_bwt_block(a[0], a[1])
        assert rc == 0 and (inv == d).all()


_SWITCH_INPUTS = []


@pytest.mark.parametrize("switch", ["BWTC_HIP_GRAMS=0", "BWTC_HIP_GRAMS=4", "BWTC_HIP_SPLIT_INDEX=0", "BWTC_HIP_PLANES=0",
                                    "BWTC_HIP_DENSE=0", "BWTC_HIP_NO_EMIT=1", "BWTC_HIP_WINDOW_BITS=12", "BWTC_HIP_SCAN=chained",
                                    # round 4: the long-key route's pieces, each switched off or cut short so that the
                                    # next piece (text rounds, late rank completion + doubling rounds) takes over
                                    "BWTC_HIP_LONG=0", "BWTC_HIP_FINISHER=0", "BWTC_HIP_FIN_PASSES=0", "BWTC_HIP_FIN_PASSES=1",
                                    "BWTC_HIP_TEXT_ROUNDS=0", "BWTC_HIP_TEXT_ROUNDS=1", "BWTC_HIP_LONG_E=6", "BWTC_HIP_LONG_G2=1",
                                    "BWTC_HIP_FIN_PASSES=0,BWTC_HIP_TEXT_ROUNDS=0",
                                    # round 5: the long key's two makers (order-1 prefix code, the default; dense gram codes),
                                    # shorter code keys, the finisher's window / largest group / characters per pass, and the
                                    # depth below which a hard group takes its rounds first (0: none does, 48: most do)
                                    "BWTC_HIP_KEYS=grams", "BWTC_HIP_KEYS=grams,BWTC_HIP_FIN_PASSES=1", "BWTC_HIP_KEYS=grams,BWTC_HIP_LONG_G2=1,BWTC_HIP_TEXT_ROUNDS=0",
                                    "BWTC_HIP_CODE_BITS=64", "BWTC_HIP_CODE_BITS=48,BWTC_HIP_FIN_PASSES=2", "BWTC_HIP_CODE_BITS=40,BWTC_HIP_TEXT_ROUNDS=1",
                                    "BWTC_HIP_FIN_FLOOR=0", "BWTC_HIP_FIN_FLOOR=48", "BWTC_HIP_FIN_FLOOR=48,BWTC_HIP_TEXT_ROUNDS=0",
                                    "BWTC_HIP_FIN_WORDS=3", "BWTC_HIP_FIN_WORDS=4,BWTC_HIP_FIN_GROUP=512",
                                    "BWTC_HIP_FIN_GROUP=1024,BWTC_HIP_FIN_WINDOW=2048", "BWTC_HIP_FIN_GROUP=512", "BWTC_HIP_FIN_WINDOW=2048,BWTC_HIP_FIN_GROUP=256",
                                    "BWTC_HIP_SPLIT_INDEX=0,BWTC_HIP_FIN_PASSES=1",
                                    # what the finisher's passes leave of deep repeats: doubled group by group in LDS beside
                                    # the global list (the default), or joined to the global list as in round 4
                                    "BWTC_HIP_LOCAL_ROUNDS=0", "BWTC_HIP_LOCAL_ROUNDS=0,BWTC_HIP_FIN_FLOOR=48", "BWTC_HIP_FIN_GROUP=512,BWTC_HIP_FIN_WORDS=3",
                                    # rounds of sixteen characters a finisher pass makes inside the workgroup (default three)
                                    "BWTC_HIP_FIN_ROUNDS=1", "BWTC_HIP_FIN_ROUNDS=2", "BWTC_HIP_FIN_ROUNDS=2,BWTC_HIP_FIN_GROUP=512", "BWTC_HIP_FIN_ROUNDS=4",
                                    "BWTC_HIP_FIN_ROUNDS=1,BWTC_HIP_FIN_PASSES=1",
                                    # the long-key sort's second-word passes with digit planes, as the key word's passes
                                    "BWTC_HIP_LONG_DIRECT=0",
                                    # a short list keeps the finisher's default shape (default: windows of 2048 entries and
                                    # groups of up to 1024 members when the long keys left at most n / 24 suffixes tied)
                                    "BWTC_HIP_FIN_WIDE=0", "BWTC_HIP_FIN_WIDE=0,BWTC_HIP_FIN_ROUNDS=2"])
def test_sorter_feature_switches_agree(oracle, monkeypatch, switch):
    """Every switch that turns a feature of the suffix sorter off (or another way) changes speed
    only: a 6 MiB text block and a block with zero bytes against the reference's sorter."""
    from bwtc_amd import hip
    for one in switch.split(","):
        name, value = one.split("=")
        monkeypatch.setenv(name, value)
    size = 6 << 20
    if not _SWITCH_INPUTS:                                    # the inputs and the reference's answers, once for all switches
        text = synth.gen_text(size, 21)
        zeros = text.copy()
        zeros[zeros == 101] = 0
        # text with repeats of every length up to 200 KB (deep ties behind the long keys: hard groups, text rounds, the bridge)
        rep = text.copy()
        rep[3 << 20:(3 << 20) + 200000] = rep[:200000]
        rep[5 << 20:(5 << 20) + 3000] = rep[1000:4000]
        for d in (text, zeros, rep):
            _SWITCH_INPUTS.append((d, oracle.oracle_bwt_block(d, 7)))
    with hip.Context(0, size) as ctx:
        for d, b in _SWITCH_INPUTS:
            a = ctx.bwt_block(d, 7)
            assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all(), switch


def test_two_contexts_in_two_threads(oracle):
    """One context per worker thread (the farm's unit): two threads transform different blocks
    on the same GPU at the same time."""
    import threading
    from bwtc_amd import hip
    blocks = [synth.gen_text(3 << 20, 3), synth.gen_dna(2 << 20, 2)]
    results = [None, None]

    def work(i):
        with hip.Context(0, 4 << 20) as ctx:
            for _ in range(3):
                results[i] = ctx.bwt_block(blocks[i], 8)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i in range(2):
        rc, inv = oracle.oracle_inverse_bwt_block(results[i][0], results[i][1])
        assert rc == 0 and (inv == blocks[i]).all()


def test_device_pointers_unaligned_and_aliased(hip_ctx, oracle):
    """bwtc_hip_bwt_block_device / _inverse_ / huffman_encode_device on caller-owned device
    buffers (as bench.py uses them), including odd source offsets and in == out."""
    d = synth.gen_text(1 << 20, 3)
    want = oracle.oracle_bwt_block(d, 8)
    buf = hip_ctx.dmalloc(d.size + 64)
    out = hip_ctx.dmalloc(hip_ctx.compress_bound(d.size))
    try:
        for off in (0, 1, 7):
            ptr = buf + off
            hip_ctx.to_device(ptr, d)
            lf, freqs = hip_ctx.bwt_block_device(ptr, ptr, d.size, 8)          # in place
            got = hip_ctx.to_host(ptr, d.size)
            assert (got == want[0]).all() and (lf == want[1]).all() and (freqs == want[2]).all(), off
            hip_ctx.inverse_bwt_block_device(ptr, ptr, d.size, lf)
            assert (hip_ctx.to_host(ptr, d.size) == d).all(), off
        hip_ctx.to_device(buf, want[0])
        n = hip_ctx.huffman_encode_device(buf, d.size, want[1], want[2], out)
        assert hip_ctx.to_host(out, n).tobytes() == oracle.oracle_huffman_encode_block(*want).tobytes()
    finally:
        hip_ctx.dfree(buf)
        hip_ctx.dfree(out)


@pytest.mark.parametrize("extra", ["", "BWTC_HIP_FIN_PASSES=1", "BWTC_HIP_TEXT_ROUNDS=1", "BWTC_HIP_FIN_PASSES=0,BWTC_HIP_TEXT_ROUNDS=0",
                                   "BWTC_HIP_KEYS=grams", "BWTC_HIP_FIN_FLOOR=48", "BWTC_HIP_CODE_BITS=48", "BWTC_HIP_FIN_ROUNDS=1",
                                   "BWTC_HIP_FIN_ROUNDS=2,BWTC_HIP_FIN_PASSES=1"])
def test_long_key_route_on_small_structured_blocks(oracle, monkeypatch, extra):
    """The long-key route (long keys, ranking at the long key's depth, finisher, text rounds, late rank completion)
    on blocks small enough for the oracle but of every shape: BWTC_HIP_GRAM_MIN_N lets blocks of a few thousand bytes
    take it.  Texts over 20-60 symbols with planted repeats of every length, periodic stretches (groups far above the
    finisher's 1024 members), runs, zero bytes, blocks that end in their own prefix."""
    from bwtc_amd import hip
    monkeypatch.setenv("BWTC_HIP_GRAM_MIN_N", "64")
    for one in extra.split(","):
        if one:
            name, value = one.split("=")
            monkeypatch.setenv(name, value)
    rng = np.random.default_rng(1234 + len(extra))
    taken = 0
    with hip.Context(0, 1 << 20) as ctx:
        for it in range(120):
            n = int(rng.integers(300, 400000))
            # alphabets of a generated text (18-64 symbols) and of a real one (150-230: round 5's code keys exist for those)
            sigma = int(rng.integers(18, 64)) if it % 4 else int(rng.integers(150, 231))
            alphabet = rng.choice(np.arange(1 if it % 3 else 0, 256), sigma, replace=False).astype(np.uint8)
            words = [alphabet[rng.integers(0, sigma, int(rng.integers(1, 9)))] for _ in range(int(rng.integers(5, 200)))]
            d = np.concatenate([words[int(i)] for i in rng.integers(0, len(words), n // 3 + 8)])[:n].copy()
            n = d.size
            shape = it % 6
            if it % 5 == 0 and n > 5000:
                # an exotic but repetitive string: symbols that occur nowhere else (the code's sample gives their pairs the
                # longest codewords: keys of one or two characters), in a run of thousands -- a hard group of little depth
                rare = np.setdiff1d(np.arange(1, 256), alphabet)[:3].astype(np.uint8)
                if rare.size == 3:
                    a = int(rng.integers(0, n - 4000))
                    d[a:a + 3000] = np.tile(rare, 1000)
            if shape == 1:                                    # planted repeats of many lengths
                for _ in range(int(rng.integers(1, 30))):
                    ln = int(min(n // 3, rng.integers(1, 1 + int(rng.choice([40, 2000, 100000])))))
                    a, b = int(rng.integers(0, n - ln)), int(rng.integers(0, n - ln))
                    d[b:b + ln] = d[a:a + ln].copy()
            elif shape == 2:                                  # a periodic stretch: one group of thousands of members
                p = d[:int(rng.integers(1, 30))].copy()
                a = int(rng.integers(0, n // 2))
                ln = int(rng.integers(2000, max(2001, n // 2)))
                d[a:a + ln] = np.tile(p, ln // p.size + 1)[:min(ln, n - a)]
            elif shape == 3:                                  # long runs of one symbol
                for _ in range(int(rng.integers(1, 8))):
                    a = int(rng.integers(0, n - 1))
                    d[a:a + int(rng.integers(1, 6000))] = alphabet[0]
            elif shape == 4:                                  # the block ends in its own beginning
                ln = int(rng.integers(1, n // 2))
                d[n - ln:] = d[:ln].copy()
            sp = int(rng.choice([1, 2, 8, 256]))
            a = ctx.bwt_block(d, sp)
            taken += 1 if ctx.stats().route & 1 else 0
            b = oracle.oracle_bwt_block(d, sp)
            assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all(), (extra, it, shape, n, sigma, sp)
    assert taken > 60, taken                                  # the route under test was the route taken


@pytest.mark.gpu
@pytest.mark.parametrize("extra", ["", "BWTC_HIP_LONG_G2=1", "BWTC_HIP_LONG_DROP=0"])
def test_long_keys_with_a_second_word_of_less_than_one_digit(oracle, monkeypatch, extra):
    """Few distinct grams make the long key's second word shorter than one radix digit (7 or 4 bits with three
    dropped): the sort then starts with the bridge digit -- the word under the key's lowest bits -- and the key
    maker has to leave exactly that digit in the first plane.  (Found by the randomised campaign: the plane held
    the word alone, the histogram and the scatter disagreed.)"""
    from bwtc_amd import hip
    monkeypatch.setenv("BWTC_HIP_GRAM_MIN_N", "64")
    if extra:
        name, value = extra.split("=")
        monkeypatch.setenv(name, value)
    rng = np.random.default_rng(77 + len(extra))
    with hip.Context(0, 1 << 20) as ctx:
        for it in range(60):
            n = int(rng.integers(200, 3000))
            sigma = int(rng.integers(3, 24))
            alphabet = rng.choice(np.arange(1 if it % 3 else 0, 256), sigma, replace=False).astype(np.uint8)
            words = [alphabet[rng.integers(0, sigma, int(rng.integers(1, 7)))] for _ in range(int(rng.integers(2, 12)))]
            d = np.concatenate([words[int(i)] for i in rng.integers(0, len(words), n + 8)])[:n].copy()
            sp = int(rng.choice([1, 8, 256]))
            a = ctx.bwt_block(d, sp)
            b = oracle.oracle_bwt_block(d, sp)
            assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all(), (extra, it, n, sigma, sp)


def test_deep_repeats_take_the_local_rounds(oracle):
    """Copies of whole files: after the finisher's passes most of what is tied sits in small groups that share thousands
    of characters.  Those double group by group in LDS (k_finish<RANK>, route bit 64) beside the global list of the
    groups too large for that; both read one rank[] and both must see each other's refinements.  16 MiB: the block's
    first 3 MiB twice more, a 40 KB stretch 60 times (groups of 60: local), a 300-byte stretch 2000 times (groups of 2000:
    global) -- against the reference's sorter."""
    from bwtc_amd import hip
    size = 16 << 20
    d = synth.gen_text(size, 77)
    d[5 << 20:8 << 20] = d[:3 << 20]
    d[11 << 20:14 << 20] = d[:3 << 20]
    for k in range(60):
        d[(9 << 20) + 40000 * k:(9 << 20) + 40000 * (k + 1)] = d[1234567:1234567 + 40000]
    for k in range(2000):
        d[(15 << 20) + 300 * k:(15 << 20) + 300 * (k + 1)] = d[7654321:7654321 + 300]
    want = oracle.ref_bwt_block(d, 8) if oracle.ref() is not None else oracle.oracle_bwt_block(d, 8)
    with hip.Context(0, size) as ctx:
        got = ctx.bwt_block(d, 8)
        st = ctx.stats()
        assert st.route & 64, st.route
        assert (got[0] == want[0]).all() and (got[1] == want[1]).all() and (got[2] == want[2]).all()
        back = ctx.inverse_bwt_block(got[0], got[1])
        assert (back == d).all()
