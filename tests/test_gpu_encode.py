"""GPU parity tests of the 'H' entropy coder front-end (C ABI -> libbwtc_hip.so) against the
oracle's restatement of HuffmanEncoder and the golden stream recorded from the reference."""
import json
import os

import numpy as np
import pytest

from bwtc_amd import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _packed(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _stream(record, size):
    # Compressor::compress framing for one precompressor block holding one BWT block
    return b"H" + _packed(size) + _packed(1) + b"\x00" + record + b"\x00"


def _inputs():
    rng = np.random.default_rng(2024)
    yield "abracadabra", np.frombuffer(b"abracadabra", np.uint8), 1
    yield "one_byte", np.frombuffer(b"x", np.uint8), 1
    yield "all_equal", np.full(70001, 7, np.uint8), 8
    yield "two_syms", np.tile(np.array([0, 255], np.uint8), 30000), 8
    yield "random_64k", rng.integers(0, 256, 65536).astype(np.uint8), 8
    yield "random_300k", rng.integers(0, 256, 300000).astype(np.uint8), 3    # ~30 sections
    yield "uniform_3M_256_sections", rng.integers(0, 256, 3 << 20).astype(np.uint8), 8
    yield "text_1M", synth.gen_text(1 << 20, 3), 8
    yield "dna_1M", synth.gen_dna(1 << 20, 2), 8
    yield "long_runs", np.repeat(rng.integers(0, 4, 3000).astype(np.uint8), rng.integers(1, 5000, 3000)), 8
    yield "skew", (rng.geometric(0.3, 500000) % 256).astype(np.uint8), 16


def test_encode_matches_oracle(hip_ctx, oracle):
    for name, data, sp in _inputs():
        bwt, lf, freqs = oracle.oracle_bwt_block(data, sp)
        want = oracle.oracle_huffman_encode_block(bwt, lf, freqs)
        got = hip_ctx.huffman_encode(bwt, lf, freqs)
        assert got.size == want.size, (name, got.size, want.size)
        assert got.tobytes() == want.tobytes(), name


def test_transform_and_encode_roundtrip(hip_ctx, oracle):
    for name, data, sp in _inputs():
        rec, bwt = hip_ctx.transform_and_encode(data, sp)
        obwt, olf, ofr = oracle.oracle_bwt_block(data, sp)
        assert (bwt == obwt).all(), name
        assert rec.tobytes() == oracle.oracle_huffman_encode_block(obwt, olf, ofr).tobytes(), name
        back = oracle.oracle_decompress_H(np.frombuffer(_stream(rec.tobytes(), data.size), np.uint8),
                                          data.size + 8)
        assert back is not None and back.tobytes() == data.tobytes(), name


def test_golden_stream_from_reference(hip_ctx):
    c = [x for x in json.load(open(os.path.join(G, "streams.json")))["cases"] if x["coder"] == "H"][0]
    data = np.frombuffer(c["input_ascii"].encode(), np.uint8)
    rec, _ = hip_ctx.transform_and_encode(data, c["sp"])
    assert _stream(rec.tobytes(), data.size) == bytes.fromhex(c["stream_hex"])


def test_encode_large_text(hip_ctx, oracle):
    size = 32 << 20
    data = synth.gen_text(size, 3)
    rec, bwt = hip_ctx.transform_and_encode(data, 8)
    back = oracle.oracle_decompress_H(np.frombuffer(_stream(rec.tobytes(), size), np.uint8), size + 8)
    assert back is not None and back.tobytes() == data.tobytes()
    print("32 MiB text -> %d bytes (ratio %.3f)" % (rec.size, rec.size / size))


def test_baseline_configs_bwt_and_H_record(hip_ctx):
    """BASELINE.json configs 1 (1 MiB random + Huffman coder) and 2 (64 MiB DNA) plus the text
    blocks: BWT bytes and the whole 'H' record against the committed checksums
    (tests/golden/bwt_large.json; BWT by the reference's divsufsort)."""
    import hashlib
    cases = json.load(open(os.path.join(G, "bwt_large.json")))["cases"]
    assert any(c["name"] == "C2_dna_64MiB" for c in cases)
    for c in cases:
        if c["size"] > (64 << 20):
            continue                      # the 256 MiB block has tests of its own (test_gpu_bwt.py)
        d = getattr(synth, c["gen"])(c["size"], c["seed"])
        rec, bwt = hip_ctx.transform_and_encode(d, c["sp"])
        assert hashlib.sha256(bwt.tobytes()).hexdigest() == c["bwt_sha256"], c["name"]
        assert rec.size == c["h_record_bytes"], c["name"]
        assert hashlib.sha256(rec.tobytes()).hexdigest() == c["h_record_sha256"], c["name"]
        # the default coder's record: the oracle's literal WaveletEncoder over the reference's BWT
        hip_ctx.wavelet_reset()
        brec, _ = hip_ctx.transform_and_encode_wavelet(d, c["sp"])
        assert brec.size == c["b_record_bytes"], c["name"]
        assert hashlib.sha256(brec.tobytes()).hexdigest() == c["b_record_sha256"], c["name"]


def test_wavelet_run_scanner_matches_oracle(hip_ctx, oracle):
    """utils::calculateRunsAndCharacters per section (what the WaveletTree constructor starts
    from): GPU front-end against the oracle's restatement, section by section."""
    rng = np.random.default_rng(31)
    long_runs = np.repeat(rng.integers(0, 6, 4000).astype(np.uint8), rng.integers(1, 9000, 4000))
    for name, data in [("text", synth.gen_text(3 << 20, 3)), ("dna", synth.gen_dna(1 << 20, 2)),
                       ("random", synth.gen_random_bytes(300000, 1)), ("long_runs", long_runs),
                       ("one_run", np.full(50000, 9, np.uint8))]:
        bwt, lf, freqs = hip_ctx.bwt_block(data, 8)
        sec, rf, tot, dist = hip_ctx.wavelet_section_stats(bwt, freqs)
        assert (sec == oracle.oracle_sections(freqs)).all(), name
        beg = 0
        for s in range(sec.size):
            orf, oruns, odist = oracle.oracle_runs_and_characters(bwt[beg:beg + int(sec[s])])
            assert (rf[s].astype(np.uint64) == orf).all(), (name, s)
            assert int(tot[s]) == oruns, (name, s)
            assert dist[s] == odist, (name, s)
            beg += int(sec[s])


def _frame(coder, rec, size):
    return coder + _packed(size) + _packed(1) + b"\x00" + rec + b"\x00"


def test_wavelet_B_record_matches_oracle(hip_ctx, oracle):
    """WaveletEncoder::transformAndEncode ('B') through the C ABI (GPU transform + run scanner,
    host tree/range coder) against the oracle's literal restatement; a fresh stream per input."""
    for name, data, sp in _inputs():
        hip_ctx.wavelet_reset()
        rec, bwt = hip_ctx.transform_and_encode_wavelet(data, sp)
        want = oracle.oracle_compress_B(data, max(data.size, 1), sp).tobytes()
        assert _frame(b"B", rec.tobytes(), data.size) == want, name


def test_wavelet_B_golden_and_multi_block_state(hip_ctx, oracle):
    c = [x for x in json.load(open(os.path.join(G, "streams.json")))["cases"] if x["coder"] == "B"][0]
    data = np.frombuffer(c["input_ascii"].encode(), np.uint8)
    hip_ctx.wavelet_reset()
    rec, _ = hip_ctx.transform_and_encode_wavelet(data, c["sp"])
    assert _frame(b"B", rec.tobytes(), data.size) == bytes.fromhex(c["stream_hex"])
    # several blocks through ONE encoder: the FSM8 state is carried from block to block
    d = synth.gen_text(700000, 3)
    bs = 250000
    hip_ctx.wavelet_reset()
    out = b"B"
    for off in range(0, d.size, bs):
        blk = d[off:off + bs]
        rec, _ = hip_ctx.transform_and_encode_wavelet(blk, 8)
        out += _packed(blk.size) + _packed(1) + b"\x00" + rec.tobytes()
    out += b"\x00"
    assert out == oracle.oracle_compress_B(d, bs, 8).tobytes()


def test_wavelet_B_host_tree_path_gives_the_same_bytes(hip_ctx, oracle):
    """BWTC_HIP_WAVELET=host builds the trees with the library's host builder instead of the
    device stream kernels (the route taken for shapes the stream path declines); both routes
    must produce the reference's bytes."""
    from bwtc_amd import hip
    os.environ["BWTC_HIP_WAVELET"] = "host"
    try:
        ctx = hip.Context(device=0, max_block_size=(4 << 20) + 1024)
    finally:
        del os.environ["BWTC_HIP_WAVELET"]
    try:
        for name, data, sp in _inputs():
            if data.size > (4 << 20):
                continue
            ctx.wavelet_reset()
            hip_ctx.wavelet_reset()
            rec_host, _ = ctx.transform_and_encode_wavelet(data, sp)
            rec_dev, _ = hip_ctx.transform_and_encode_wavelet(data, sp)
            assert rec_host.tobytes() == rec_dev.tobytes(), name
    finally:
        ctx.close()


def test_wavelet_B_large_text_block(hip_ctx, oracle):
    """32 MiB of text: tens of millions of steps through the stream kernels; whole record
    against the oracle (a few seconds of CPU)."""
    data = synth.gen_text(32 << 20, 5)
    hip_ctx.wavelet_reset()
    rec, bwt = hip_ctx.transform_and_encode_wavelet(data, 8)
    want = oracle.oracle_compress_B(data, data.size, 8).tobytes()
    assert _frame(b"B", rec.tobytes(), data.size) == want
    # and the other extreme of the run structure: DNA, almost every byte its own run, 4 symbols
    data = synth.gen_dna(16 << 20, 9)
    hip_ctx.wavelet_reset()
    rec, bwt = hip_ctx.transform_and_encode_wavelet(data, 8)
    assert _frame(b"B", rec.tobytes(), data.size) == oracle.oracle_compress_B(data, data.size, 8).tobytes()


def test_wavelet_B_overlapped_blocks_equal_the_sequential_stream(hip_ctx, oracle):
    """Several blocks through _begin/_end with all of them under way at once: the carried model
    state is handed on at _begin, so the records are those of the sequential encoder."""
    d = synth.gen_text(6 << 20, 9)
    bs = 1 << 20
    hip_ctx.wavelet_reset()
    bufs, tickets, sizes = [], [], []
    for off in range(0, d.size, bs):
        blk = d[off:off + bs]
        d_in = hip_ctx.dmalloc(blk.size + 16)
        try:
            hip_ctx.to_device(d_in, blk)
            lf, freqs = hip_ctx.bwt_block_device(d_in, d_in, blk.size, 8)
            out = np.zeros(hip_ctx.compress_bound(blk.size), np.uint8)
            tickets.append(hip_ctx.wavelet_encode_device_begin(d_in, blk.size, lf, freqs, out, threads=4))
        finally:
            hip_ctx.dfree(d_in)         # the block's streams have left the device by now
        bufs.append(out)
        sizes.append(blk.size)
    stream = b"B"
    for t, out, n in zip(tickets, bufs, sizes):
        m = hip_ctx.wavelet_encode_end(t)
        stream += _packed(n) + _packed(1) + b"\x00" + out[:m].tobytes()
    stream += b"\x00"
    assert stream == oracle.oracle_compress_B(d, bs, 8).tobytes()


def test_wavelet_B_short_structured_streams_through_the_pipeline(hip_ctx, oracle):
    """Twenty-four short streams (all-equal bytes, periods with defects, Fibonacci words, tiny
    alphabets, repeats, text, DNA; tests/blockgen.py) cut into blocks of random size, up to twelve
    under way at once through _begin/_end: block sizes from a few hundred bytes up change the route
    from block to block (host tree builder / device streams, models on the device), buffers are
    recycled across sizes, and the carried state runs through all of it.  Each stream byte-equal to
    the oracle's sequential encoder.  (The fixed-seed part of scripts/fuzz_gpu_parity.py.)"""
    import blockgen
    rng = np.random.default_rng(424242)
    for case in range(24):
        total = int(rng.integers(1000, 3 << 20))
        kind, d = blockgen.structured(rng, total)
        bs = int(rng.integers(max(200, total // 40), max(400, total)))
        hip_ctx.wavelet_reset()
        pending, records = [], []

        def collect():
            t, out, n = pending.pop(0)
            m = hip_ctx.wavelet_encode_end(t)
            records.append(_packed(n) + _packed(1) + b"\x00" + out[:m].tobytes())
        for off in range(0, d.size, bs):
            blk = d[off:off + bs]
            if len(pending) >= 12:
                collect()
            d_in = hip_ctx.dmalloc(blk.size + 16)
            try:
                hip_ctx.to_device(d_in, blk)
                lf, freqs = hip_ctx.bwt_block_device(d_in, d_in, blk.size, 8)
                out = np.zeros(hip_ctx.compress_bound(blk.size), np.uint8)
                pending.append((hip_ctx.wavelet_encode_device_begin(d_in, blk.size, lf, freqs, out, threads=4), out, blk.size))
            finally:
                hip_ctx.dfree(d_in)
        while pending:
            collect()
        stream = b"B" + b"".join(records) + b"\x00"
        assert stream == oracle.oracle_compress_B(d, bs, 8).tobytes(), (case, kind, total, bs)


@pytest.mark.parametrize("models", ["device", "host"])
def test_wavelet_B_prepare_queue_flow_on_one_context(models, oracle, monkeypatch):
    """bwtc_hip_wavelet_encode_device_prepare / _queue as a caller that farms a stream uses them
    (bwtc_hip_farm.hpp), here on one context: the carried state goes in at _queue and comes back
    at once (the device passes of a prepared block leave it for all eight start states).  Two
    blocks prepared before the first is queued: the earlier one loses its turn in the device
    workspace and is modelled by the worker threads -- same stream.  Byte-equal to the oracle's
    sequential encoder, with the models on the device and on the host."""
    from bwtc_amd import hip
    if models == "host":
        monkeypatch.setenv("BWTC_HIP_MODELS", "host")
    d = np.concatenate([synth.gen_text(3 << 20, 21), synth.gen_dna(1 << 20, 5), synth.gen_text(1 << 20, 22)])
    bs = 1 << 20
    ctx = hip.Context(device=0, max_block_size=bs + 1024)
    try:
        blocks = [d[o:o + bs] for o in range(0, d.size, bs)]
        prepared = []

        def prepare(blk):
            d_in = ctx.dmalloc(blk.size + 16)
            try:
                ctx.to_device(d_in, blk)
                lf, freqs = ctx.bwt_block_device(d_in, d_in, blk.size, 8)
                out = np.zeros(ctx.compress_bound(blk.size), np.uint8)
                prepared.append((ctx.wavelet_encode_device_prepare(d_in, blk.size, lf, freqs, out, threads=4), out, blk.size))
            finally:
                ctx.dfree(d_in)
        state, joined, stream = 4, 0, b"B"                # FSM8's state in a new stream
        order = [2, 1, 1, 1]                              # blocks prepared before the next _queue calls
        at = 0
        for burst in order:
            for _ in range(burst):
                prepare(blocks[at]); at += 1
            while joined < len(prepared):
                state = ctx.wavelet_encode_queue(prepared[joined][0], state)
                joined += 1
        assert at == len(blocks)
        for t, out, n in prepared:
            m = ctx.wavelet_encode_end(t)
            stream += _packed(n) + _packed(1) + b"\x00" + out[:m].tobytes()
        stream += b"\x00"
        assert stream == oracle.oracle_compress_B(d, bs, 8).tobytes()
    finally:
        ctx.close()


def test_wavelet_B_pipeline_limits_and_misuse(oracle):
    """Depth limit and ticket handling of _begin/_end: an unknown ticket is refused, a finished
    but uncollected oldest block makes a further _begin return -6 instead of waiting forever,
    and collecting in any order still gives every block's own record."""
    from bwtc_amd import hip
    os.environ["BWTC_HIP_WAVELET_DEPTH"] = "2"
    try:
        ctx = hip.Context(device=0, max_block_size=(1 << 20) + 1024)
    finally:
        del os.environ["BWTC_HIP_WAVELET_DEPTH"]
    try:
        blocks = [synth.gen_text(300000, 40 + i) for i in range(3)]
        outs, tickets, metas = [], [], []
        d_in = ctx.dmalloc((1 << 20) + 64)
        ctx.wavelet_reset()
        for blk in blocks[:2]:
            ctx.to_device(d_in, blk)
            lf, freqs = ctx.bwt_block_device(d_in, d_in, blk.size, 8)
            out = np.zeros(ctx.compress_bound(blk.size), np.uint8)
            tickets.append(ctx.wavelet_encode_device_begin(d_in, blk.size, lf, freqs, out, threads=2))
            outs.append(out)
        with pytest.raises(hip.BwtcHipError):
            ctx.wavelet_encode_end(987654)                       # never issued
        # wait until the oldest is done without collecting it, then a third _begin must say -6
        import time
        time.sleep(2.0)
        ctx.to_device(d_in, blocks[2])
        lf, freqs = ctx.bwt_block_device(d_in, d_in, blocks[2].size, 8)
        out3 = np.zeros(ctx.compress_bound(blocks[2].size), np.uint8)
        with pytest.raises(hip.BwtcHipError) as err:
            ctx.wavelet_encode_device_begin(d_in, blocks[2].size, lf, freqs, out3, threads=2)
        assert "-6" in str(err.value)
        # collect newest first
        n1 = ctx.wavelet_encode_end(tickets[1])
        n0 = ctx.wavelet_encode_end(tickets[0])
        with pytest.raises(hip.BwtcHipError):
            ctx.wavelet_encode_end(tickets[0])                   # already collected
        stream = b"B"
        for blk, out, n in zip(blocks[:2], outs, (n0, n1)):
            stream += _packed(blk.size) + _packed(1) + b"\x00" + out[:n].tobytes()
        stream += b"\x00"
        want = oracle.oracle_compress_B(np.concatenate(blocks[:2]), 300000, 8).tobytes()
        assert stream == want
        ctx.dfree(d_in)
    finally:
        ctx.close()


def test_wavelet_other_model_letters_on_the_device_route(hip_ctx, oracle):
    """Coder letters 'b' and 'u' through the C ABI (bwtc_hip_wavelet_start picks the main model;
    the stream kernels are the same, only the host models differ) against the oracle, two blocks
    per stream; 'm' / 'M' are refused."""
    from bwtc_amd import hip
    d = synth.gen_text(900000, 17)
    bs = 500000
    try:
        for coder in ("b", "u"):
            hip_ctx.wavelet_start(coder)
            out = coder.encode()
            for off in range(0, d.size, bs):
                blk = d[off:off + bs]
                rec, _ = hip_ctx.transform_and_encode_wavelet(blk, 8)
                out += _packed(blk.size) + _packed(1) + b"\x00" + rec.tobytes()
            out += b"\x00"
            assert out == oracle.oracle_compress_wavelet(coder, d, bs, 8).tobytes(), coder
        for coder in ("m", "M", "H", "x"):
            with pytest.raises(hip.BwtcHipError):
                hip_ctx.wavelet_start(coder)
    finally:
        hip_ctx.wavelet_reset()


@pytest.mark.parametrize("models", ["device", "host", "fallback"])
def test_wavelet_B_models_on_the_device_and_on_the_host_give_the_same_stream(models, oracle, monkeypatch):
    """The adaptive models run on the GPU (wavelet_gpu_models.hip: state scan, slot space, bracketed
    chains; the host only range-codes) or on the worker threads (BWTC_HIP_MODELS=host): either way
    the sequential encoder's bytes, over a stream of blocks of different kinds whose model state
    runs on from block to block, and with the scans done by the chained single-launch kernel."""
    from bwtc_amd import hip
    if models == "host":
        monkeypatch.setenv("BWTC_HIP_MODELS", "host")
        monkeypatch.setenv("BWTC_HIP_DENSE_STEP_KEYS", "1")      # and the opt-in dense sort ids of the steps
    elif models == "fallback":
        # the device passes run, then every block is treated as flagged: its models are redone on the
        # worker threads from the packed streams (the route that keeps a surprise from costing bytes)
        monkeypatch.setenv("BWTC_HIP_TEST_MODELS_FALLBACK", "1")
        monkeypatch.setenv("BWTC_HIP_SEG_STEPS", "0")            # and the step sort over block-wide group numbers
    else:
        monkeypatch.setenv("BWTC_HIP_SCAN", "chained")
        monkeypatch.setenv("BWTC_HIP_GM_PARTITION", "lines")     # and the partition pass that stores whole lines
    rng = np.random.default_rng(31)
    parts = [synth.gen_text(3 << 20, 5), synth.gen_dna(1 << 20, 6), np.full(300000, 9, np.uint8),
             rng.integers(0, 256, 700000).astype(np.uint8), synth.gen_text((2 << 20) + 12345, 7),
             np.tile(np.array([0, 1], np.uint8), 400000)]
    bs = 3 << 20
    d = np.concatenate(parts)
    with hip.Context(0, bs) as ctx:
        ctx.wavelet_reset()
        pend, stream = [], b"B"
        for off in range(0, d.size, bs):
            blk = d[off:off + bs]
            d_in = ctx.dmalloc(blk.size + 16)
            try:
                ctx.to_device(d_in, blk)
                lf, freqs = ctx.bwt_block_device(d_in, d_in, blk.size, 8)
                out = np.zeros(ctx.compress_bound(blk.size), np.uint8)
                pend.append((ctx.wavelet_encode_device_begin(d_in, blk.size, lf, freqs, out, threads=4), out, blk.size))
            finally:
                ctx.dfree(d_in)
        for t, out, n in pend:
            stream += _packed(n) + _packed(1) + b"\x00" + out[:ctx.wavelet_encode_end(t)].tobytes()
        stream += b"\x00"
    assert stream == oracle.oracle_compress_B(d, bs, 8).tobytes(), models


def test_wavelet_start_with_begun_blocks_not_yet_in_their_stream(hip_ctx, oracle):
    """_begin lets a block join its stream one or two calls later.  A new stream started in
    between (bwtc_hip_wavelet_start) must not reach back: the begun blocks are coded with the old
    stream's state and model, the next block starts the new one."""
    d = synth.gen_text(3 * 400000, 23)
    bs = 400000
    blocks = [d[o:o + bs] for o in range(0, d.size, bs)]

    def begin(blk):
        d_in = hip_ctx.dmalloc(blk.size + 16)
        try:
            hip_ctx.to_device(d_in, blk)
            lf, freqs = hip_ctx.bwt_block_device(d_in, d_in, blk.size, 8)
            out = np.zeros(hip_ctx.compress_bound(blk.size), np.uint8)
            return hip_ctx.wavelet_encode_device_begin(d_in, blk.size, lf, freqs, out, threads=4), out
        finally:
            hip_ctx.dfree(d_in)

    try:
        hip_ctx.wavelet_reset()
        a = [begin(blocks[0]), begin(blocks[1])]          # stream 1 ('B'): two blocks begun, none collected
        hip_ctx.wavelet_start("b")                        # stream 2 starts here
        b = [begin(blocks[2])]
        first = b"B"
        for (t, out), blk in zip(a, blocks[:2]):
            first += _packed(blk.size) + _packed(1) + b"\x00" + out[:hip_ctx.wavelet_encode_end(t)].tobytes()
        first += b"\x00"
        second = b"b"
        for (t, out), blk in zip(b, blocks[2:]):
            second += _packed(blk.size) + _packed(1) + b"\x00" + out[:hip_ctx.wavelet_encode_end(t)].tobytes()
        second += b"\x00"
        assert first == oracle.oracle_compress_wavelet("B", d[:2 * bs], bs, 8).tobytes()
        assert second == oracle.oracle_compress_wavelet("b", d[2 * bs:], bs, 8).tobytes()
    finally:
        hip_ctx.wavelet_reset()


def test_wavelet_B_context_closed_with_blocks_under_way():
    """bwtc_hip_destroy with uncollected blocks: they are finished (their records land in the
    buffers given to _begin), nothing hangs or crashes."""
    from bwtc_amd import hip
    ctx = hip.Context(device=0, max_block_size=(1 << 20) + 1024)
    outs = []
    d_in = ctx.dmalloc((1 << 20) + 64)
    ctx.wavelet_reset()
    for i in range(3):
        blk = synth.gen_text(400000, 70 + i)
        ctx.to_device(d_in, blk)
        lf, freqs = ctx.bwt_block_device(d_in, d_in, blk.size, 8)
        out = np.zeros(ctx.compress_bound(blk.size), np.uint8)
        ctx.wavelet_encode_device_begin(d_in, blk.size, lf, freqs, out, threads=3)
        outs.append(out)
    ctx.dfree(d_in)
    ctx.close()
    for out in outs:
        n = int.from_bytes(out[:6].tobytes(), "big")
        assert 1000 < n < out.size and out[6 + n - 1] == 255     # closed record: length patched, coder flushed


def test_varied_block_sweep_both_coders(hip_ctx, oracle):
    """150 small blocks of varied shape (tests/blockgen.py) through the device halves of both
    coders, each against the oracle: rare shapes of the section / tree / run-length machinery."""
    import blockgen
    lf = np.zeros(1, np.uint32)
    for case, kind, bwt in blockgen.varied_blocks(150, 60000, seed=777):
        freqs = np.bincount(bwt, minlength=256).astype(np.uint32)   # any byte string serves as a transformed block
        hip_ctx.wavelet_reset()
        got = hip_ctx.wavelet_encode(bwt, lf, freqs, threads=2)
        want = oracle.oracle_wavelet_encode_block(bwt, lf, freqs)
        assert got.tobytes() == want.tobytes(), ("B", case, kind, bwt.size)
        got = hip_ctx.huffman_encode(bwt, lf, freqs)
        want = oracle.oracle_huffman_encode_block(bwt, lf, freqs)
        assert got.tobytes() == want.tobytes(), ("H", case, kind, bwt.size)


def test_wavelet_B_huge_group_tasks_at_test_size(oracle):
    """The scalar tasks for huge groups (two groups stepped alternately) only appear at 256 MiB
    with the default threshold; lowered to 1 Mi elements they run on a 32 MiB block, whose record
    is checked against the oracle."""
    from bwtc_amd import hip
    os.environ["BWTC_HIP_HUGE_MI"] = "1"
    try:
        ctx = hip.Context(device=0, max_block_size=(32 << 20) + 1024)
    finally:
        del os.environ["BWTC_HIP_HUGE_MI"]
    try:
        data = synth.gen_text(32 << 20, 5)
        ctx.wavelet_reset()
        rec, _ = ctx.transform_and_encode_wavelet(data, 8)
        assert _frame(b"B", rec.tobytes(), data.size) == oracle.oracle_compress_B(data, data.size, 8).tobytes()
    finally:
        ctx.close()


def test_wavelet_streams_cross_decode_with_the_oracle_decoder(hip_ctx, oracle):
    """test/CompressorAndDecompressorTest.cpp:61-160 run across the boundary: streams written by
    the product (GPU transform + stream kernels + host models / range coder) are decoded by the
    oracle's LITERAL restatement of the reference decoder (readShape, decodeTreeBF, message,
    BitDecoder) -- sizes 100..100 000, repetitions 0 / 2 / 50, one and several blocks, 1..30
    starting points.  (The other direction -- oracle-encoded, product-decoded -- is
    tests/cpp/wavelet_decoder_test.cpp in the CPU suite.)"""
    rng = np.random.default_rng(99)

    def make(length, reps):
        if reps == 0:
            return rng.integers(0, 256, length, dtype=np.uint8)
        return np.tile(rng.integers(0, 256, length // reps, dtype=np.uint8), reps)

    cases = []
    for length in (100, 1000, 10000, 100000):
        for reps in (0, 2, 50):
            cases.append((length, reps, int(length * 100 * 0.185), 1))
            cases.append((length, reps, max(int(length * (10 if reps != 2 else 1) * 0.185), 18), 1))
    for sp in range(1, 31):
        cases.append((10000, 0, int(100000 * 0.185), sp))
    for length, reps, block, sp in cases:
        d = make(length, reps)
        hip_ctx.wavelet_reset()
        out = b"B"
        for off in range(0, d.size, block):
            blk = d[off:off + block]
            rec, _ = hip_ctx.transform_and_encode_wavelet(blk, sp)
            out += _packed(blk.size) + _packed(1) + b"\x00" + rec.tobytes()
        out += b"\x00"
        back = oracle.oracle_decompress_wavelet(np.frombuffer(out, np.uint8), d.size + 8)
        assert back is not None and back.tobytes() == d.tobytes(), (length, reps, block, sp)
        assert out == oracle.oracle_compress_B(d, block, sp).tobytes(), (length, reps, block, sp)


@pytest.mark.parametrize("switch", ["BWTC_HIP_CODER_LANES", "BWTC_HIP_FUSED"])
def test_wavelet_B_lane_engines_give_the_sequential_stream(switch, oracle):
    """The two host routes that deep pipelines use -- the 16-lane range-coder engines over stored
    probabilities, and the fused model + coder engines -- forced on for a short pipeline: eight
    blocks under way at once must still give the oracle's sequential stream (with the fused
    engines the first blocks of the stream take the two-stage route, the later ones the lanes)."""
    from bwtc_amd import hip
    os.environ[switch] = "1"
    try:
        os.environ["BWTC_HIP_WAVELET_DEPTH"] = "32"
        ctx = hip.Context(device=0, max_block_size=(2 << 20) + 1024)
    finally:
        del os.environ[switch]
        del os.environ["BWTC_HIP_WAVELET_DEPTH"]
    try:
        d = np.concatenate([synth.gen_text(9 << 20, 21), synth.gen_dna(2 << 20, 5), synth.gen_random_bytes(1 << 20, 9),
                            synth.gen_text(7 << 20, 22)])
        bs = 1 << 20
        ctx.wavelet_reset()
        d_in = ctx.dmalloc(bs + 64)
        outs, tickets, sizes = [], [], []
        for off in range(0, d.size, bs):
            blk = d[off:off + bs]
            ctx.to_device(d_in, blk)
            lf, freqs = ctx.bwt_block_device(d_in, d_in, blk.size, 8)
            out = np.zeros(ctx.compress_bound(blk.size), np.uint8)
            tickets.append(ctx.wavelet_encode_device_begin(d_in, blk.size, lf, freqs, out, threads=6))
            outs.append(out)
            sizes.append(blk.size)
        ctx.dfree(d_in)
        stream = b"B"
        for t, out, n in zip(tickets, outs, sizes):
            m = ctx.wavelet_encode_end(t)
            stream += _packed(n) + _packed(1) + b"\x00" + out[:m].tobytes()
        stream += b"\x00"
        assert stream == oracle.oracle_compress_B(d, bs, 8).tobytes(), switch
    finally:
        ctx.close()
