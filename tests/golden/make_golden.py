#!/usr/bin/env python3
"""Generates tests/golden/bwt_small.json from the reference itself.

Run in the build container only (it needs /root/reference): the reference's own, unmodified
libdivsufsort is compiled by oracle/Makefile into oracle/_ref/libbwtc_ref.so and driven
through oracle/ref_driver.c.  The output holds inputs and expected outputs only (data, not
source).  Cases mirror what the reference's tests exercise: random blocks of several alphabet
sizes incl. byte 0 (test/InverseBwtTest.cpp:51-114), repetitive data
(test/CompressorAndDecompressorTest.cpp:52-59), size <= 256 forcing one LF power
(BWTBlock.cpp:104-108) and starting points 1..300 (test/LFpowersTest.cpp:112-114).
"""
import base64
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib  # noqa: E402
from bwtc_amd import synth  # noqa: E402


def main():
    oracle_lib.build()
    assert oracle_lib.ref() is not None, "reference build missing"
    rng = np.random.default_rng(20261003)
    cases = []

    def add(name, data, sp):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        bwt, lf, freqs = oracle_lib.ref_bwt_block(data, sp)
        cases.append({
            "name": name, "sp": sp, "size": int(data.size),
            "input": base64.b64encode(data.tobytes()).decode(),
            "bwt": base64.b64encode(bwt.tobytes()).decode(),
            "lf": [int(x) for x in lf],
            "freqs_nonzero": {str(i): int(c) for i, c in enumerate(freqs) if c},
        })

    for s in [b"a", b"ab", b"ba", b"aa", b"abracadabra", b"mississippi", b"banana", b"aaaaaaaaaa",
              b"\x00", b"\x00\x00\x00", b"\x00\x00\x00\x00\x00", b"\x01\x00\x00\x00", b"abab\x00\x00"]:
        add("lit_%s" % s.hex(), np.frombuffer(s, np.uint8), 1)
    add("abra300_sp3", np.frombuffer((b"abracadabra" * 28)[:300], np.uint8), 3)
    for i in range(36):
        n = int(rng.integers(1, 4097))
        sigma = int(rng.choice([1, 2, 4, 256]))
        d = rng.integers(0, sigma, n).astype(np.uint8)
        kind = i % 4
        if kind == 1:
            d = np.tile(d[:max(1, n // 9)], 10)[:n]          # periodic
        elif kind == 2 and sigma > 1:
            d[rng.integers(0, n, n // 3)] = 0                # plenty of zero bytes
        sp = int(rng.choice([1, 2, 8, 256, 300]))
        add("rand%02d_n%d_s%d_k%d" % (i, n, sigma, kind), d, sp)
    add("text4k", synth.gen_text(4000, 3), 8)
    add("dna4k", synth.gen_dna(4000, 2), 8)
    add("size256", rng.integers(0, 256, 256).astype(np.uint8), 8)
    add("size257", rng.integers(0, 256, 257).astype(np.uint8), 8)
    if "--large-only" not in sys.argv:
        with open(os.path.join(HERE, "bwt_small.json"), "w") as f:
            json.dump({"generator": "tests/golden/make_golden.py (reference libdivsufsort via oracle/_ref)",
                       "cases": cases}, f, indent=0)
        print("wrote %d cases" % len(cases))

    # Larger blocks: checksum only.
    import hashlib
    big = []
    for name, gen, size, seed, sp in [("C1_random_1MiB", synth.gen_random_bytes, 1 << 20, 1, 8),
                                      ("C2_dna_4MiB", synth.gen_dna, 4 << 20, 2, 8),
                                      ("C3_text_4MiB", synth.gen_text, 4 << 20, 3, 8),
                                      ("C3_text_16MiB", synth.gen_text, 16 << 20, 3, 8),
                                      ("C2_dna_64MiB", synth.gen_dna, 64 << 20, 2, 8),
                                      ("C3_text_256MiB", synth.gen_text, 256 << 20, 3, 8)]:
        d = gen(size, seed)
        bwt, lf, freqs = oracle_lib.ref_bwt_block(d, sp)
        # 'H' record of the block: the oracle's encoder (pinned by the reference's golden
        # stream in streams.json) over the reference's BWT output
        rec = oracle_lib.oracle_huffman_encode_block(bwt, lf, freqs)
        # 'B' record (the default coder, BASELINE config 3): the oracle's literal WaveletEncoder
        # over the reference's BWT output, a fresh encoder per block
        brec = oracle_lib.oracle_wavelet_encode_block(bwt, lf, freqs)
        big.append({"name": name, "gen": gen.__name__, "size": size, "seed": seed, "sp": sp,
                    "b_record_bytes": int(brec.size),
                    "b_record_sha256": hashlib.sha256(brec.tobytes()).hexdigest(),
                    "input_sha256": hashlib.sha256(d.tobytes()).hexdigest(),
                    "bwt_sha256": hashlib.sha256(bwt.tobytes()).hexdigest(),
                    "lf": [int(x) for x in lf],
                    "h_record_bytes": int(rec.size),
                    "h_record_sha256": hashlib.sha256(rec.tobytes()).hexdigest()})
        print(name, big[-1]["bwt_sha256"][:16], big[-1]["lf"][:3])
    with open(os.path.join(HERE, "bwt_large.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py (reference libdivsufsort via oracle/_ref)",
                   "cases": big}, f, indent=1)


if __name__ == "__main__":
    main()
