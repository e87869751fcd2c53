#!/usr/bin/env python3
"""Time-bounded randomised parity campaign on the GPU box (not part of the test suite: the suite's
sweeps are the fixed-seed subset of this).  Three phases, each against the oracle:
  1. byte strings of varied shape as transformed blocks through both coders' device halves;
  2. multi-block 'B' streams with random block sizes through the overlapped _begin/_end pipeline
     (carried model state, models on the device);
  3. the transform itself on repetitive / periodic / small-alphabet inputs;
  4. the GPU inverse transform of the product's own transform (gives the input back);
  5. `compress` -> `uncompress` round trips of the command-line tools (host decoders + GPU inverse),
     both coders, and the 'B' / 'H' stream against the oracle's whole-stream encoder;
  6. (round 4) the long-key route -- long keys, finisher, text rounds, late rank completion, each cut short in turn -- on
     small blocks of every shape, and `--prepr` through the device sweeps.
usage: fuzz_gpu_parity.py [seconds per phase] [seed] [phases, e.g. 23]      exit code 1 if anything differed.
FUZZ_BIG=1: phase 2 with streams of 32-96 MB in blocks of 9-33 MB (the page-locked buffers of such blocks are
registered mappings, recycled across sizes)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import blockgen      # noqa: E402
import oracle_lib    # noqa: E402  (the checker)
from bwtc_amd import hip, synth   # noqa: E402


def packed(v):
    b = bytearray()
    while True:
        x = v & 0x7F
        v >>= 7
        b.append(x | (0x80 if v else 0))
        if not v:
            return bytes(b)


FUZZ6_KEYS = ("BWTC_HIP_FIN_PASSES", "BWTC_HIP_TEXT_ROUNDS", "BWTC_HIP_LONG_G2", "BWTC_HIP_FINISHER", "BWTC_HIP_FIN_GROUP", "BWTC_HIP_KEYS",
              "BWTC_HIP_CODE_BITS", "BWTC_HIP_FIN_FLOOR", "BWTC_HIP_LOCAL_ROUNDS", "BWTC_HIP_FIN_WINDOW", "BWTC_HIP_FIN_WORDS", "BWTC_HIP_FIN_ROUNDS", "BWTC_HIP_FIN_WIDE")


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    phases = sys.argv[3] if len(sys.argv) > 3 else "123456"
    oracle_lib.build()
    rng = np.random.default_rng(seed)
    big = os.environ.get("FUZZ_BIG") == "1"
    ctx = hip.Context(device=0, max_block_size=((40 if big else 8) << 20) + 1024)
    bad = 0
    print("seed", seed, flush=True)

    # ---- 1: coders on byte strings of varied shape ---------------------------------------------
    t0, n1 = time.time(), 0
    lf = np.zeros(1, np.uint32)
    while "1" in phases and time.time() - t0 < budget:
        s = int(rng.integers(1, 1 << 30))
        top = int(rng.choice([2000, 60000, 600000, 2500000]))
        for case, kind, bwt in blockgen.varied_blocks(8, top, seed=s):
            freqs = np.bincount(bwt, minlength=256).astype(np.uint32)
            ctx.wavelet_reset()
            if ctx.wavelet_encode(bwt, lf, freqs, threads=4).tobytes() != oracle_lib.oracle_wavelet_encode_block(bwt, lf, freqs).tobytes():
                bad += 1
                print("MISMATCH phase 1 'B': blockgen seed", s, "top", top, "case", case, "kind", kind, "size", bwt.size, flush=True)
            if ctx.huffman_encode(bwt, lf, freqs).tobytes() != oracle_lib.oracle_huffman_encode_block(bwt, lf, freqs).tobytes():
                bad += 1
                print("MISMATCH phase 1 'H': blockgen seed", s, "top", top, "case", case, "kind", kind, "size", bwt.size, flush=True)
            n1 += 1
    print("phase 1: %d blocks through both coders" % n1, flush=True)

    # ---- 2: multi-block 'B' streams through the overlapped pipeline ---------------------------------
    t0, n2 = time.time(), 0
    while "2" in phases and time.time() - t0 < budget:
        total = int(rng.integers(32 << 20, 96 << 20)) if big else int(rng.integers(1000, 6 << 20))
        kind, d = blockgen.structured(rng, total)
        bs = int(rng.integers(9 << 20, 33 << 20)) if big else int(rng.integers(max(200, total // 40), max(400, total)))
        print("stream", n2, "kind", kind, "total", total, "block", bs, flush=True)
        ctx.wavelet_reset()
        pending, records = [], []
        def collect():
            t, out, n = pending.pop(0)
            m = ctx.wavelet_encode_end(t)
            records.append(packed(n) + packed(1) + b"\x00" + out[:m].tobytes())
        for off in range(0, d.size, bs):
            blk = d[off:off + bs]
            if len(pending) >= 12:                       # the context keeps at most 16 blocks under way
                collect()
            d_in = ctx.dmalloc(blk.size + 16)
            try:
                ctx.to_device(d_in, blk)
                lfp, freqs = ctx.bwt_block_device(d_in, d_in, blk.size, 8)
                out = np.zeros(ctx.compress_bound(blk.size), np.uint8)
                pending.append((ctx.wavelet_encode_device_begin(d_in, blk.size, lfp, freqs, out, threads=4), out, blk.size))
            finally:
                ctx.dfree(d_in)
        while pending:
            collect()
        stream = b"B" + b"".join(records) + b"\x00"
        if stream != oracle_lib.oracle_compress_B(d, bs, 8).tobytes():
            bad += 1
            print("MISMATCH phase 2: kind", kind, "total", total, "block", bs, "(campaign seed %d, stream %d)" % (seed, n2), flush=True)
        n2 += 1
    print("phase 2: %d streams" % n2, flush=True)

    # ---- 3: the transform on structured inputs --------------------------------------------------
    t0, n3 = time.time(), 0
    while "3" in phases and time.time() - t0 < budget:
        n = int(rng.integers(1, int(rng.choice([300, 70000, 3 << 20]))))
        kind, d = blockgen.structured(rng, n)
        sp = int(rng.choice([1, 2, 8, 37, 256]))
        print("transform", n3, "kind", kind, "size", n, "sp", sp, flush=True)
        got = ctx.bwt_block(d, sp)
        want = oracle_lib.oracle_bwt_block(d, sp)
        if not all((np.asarray(g) == np.asarray(w)).all() for g, w in zip(got, want)):
            bad += 1
            print("MISMATCH phase 3: kind", kind, "size", n, "starting points", sp, "(campaign seed %d, case %d)" % (seed, n3), flush=True)
        n3 += 1
    print("phase 3: %d transforms" % n3, flush=True)
    # ---- 4: inverse transform ------------------------------------------------------------------
    t0, n4 = time.time(), 0
    while "4" in phases and time.time() - t0 < budget:
        n = int(rng.integers(1, int(rng.choice([300, 70000, 3 << 20]))))
        kind, d = blockgen.structured(rng, n)
        sp = int(rng.choice([1, 2, 8, 37, 256]))
        print("inverse", n4, "kind", kind, "size", n, "sp", sp, flush=True)
        bwt, lfp, _ = ctx.bwt_block(d, sp)
        back = ctx.inverse_bwt_block(bwt, lfp)
        if np.asarray(back).tobytes() != d.tobytes():
            bad += 1
            print("MISMATCH phase 4: kind", kind, "size", n, "starting points", sp, flush=True)
        n4 += 1
    print("phase 4: %d inverse transforms" % n4, flush=True)
    ctx.close()

    # ---- 5: command-line round trips ---------------------------------------------------------------
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "bwtc_amd", "host", "compress")
    unexe = os.path.join(ROOT, "bwtc_amd", "host", "uncompress")
    t0, n5 = time.time(), 0
    with tempfile.TemporaryDirectory() as tmp:
        while "5" in phases and time.time() - t0 < budget:
            n = int(rng.integers(1, int(rng.choice([5000, 400000, 2 << 20]))))
            kind, d = blockgen.structured(rng, n)
            enc = "B" if n5 % 2 == 0 else "H"
            print("cli", n5, "kind", kind, "size", n, "coder", enc, "two contexts" if n5 % 3 == 2 else "", flush=True)
            src, dst, back = os.path.join(tmp, "in"), os.path.join(tmp, "c"), os.path.join(tmp, "out")
            open(src, "wb").write(d.tobytes())
            farm = ["--devices", "0,0"] if n5 % 3 == 2 else []        # every third stream over two contexts on the device
            r = subprocess.run([exe, "-m", "1", "-e", enc] + farm + [src, dst], capture_output=True, text=True, timeout=300)
            ok = r.returncode == 0
            if ok:
                r = subprocess.run([unexe, dst, back], capture_output=True, text=True, timeout=300)
                ok = r.returncode == 0 and open(back, "rb").read() == d.tobytes()
            if ok:
                want = (oracle_lib.oracle_compress_B if enc == "B" else oracle_lib.oracle_compress_H)(d, 185000, 8).tobytes()
                ok = open(dst, "rb").read() == want
            if not ok:
                bad += 1
                print("MISMATCH phase 5: kind", kind, "size", n, "coder", enc, (r.stderr or "")[-300:], flush=True)
            n5 += 1
    print("phase 5: %d round trips" % n5, flush=True)

    # ---- 6: the long-key route on small blocks of every shape (round 4) -----------------------------
    # BWTC_HIP_GRAM_MIN_N lets blocks of a few thousand bytes take long keys, finisher, text rounds and the late
    # rank completion; every few blocks a new context with one of the route's pieces cut short.  Also `--prepr`
    # through the device sweeps.
    t0, n6 = time.time(), 0
    # (round 5: code keys are the default; gram keys, shorter code keys, the shallow floor, the local rounds off, the
    # finisher's window / group / width as variants)
    variants = [{}, {"BWTC_HIP_FIN_PASSES": "1"}, {"BWTC_HIP_TEXT_ROUNDS": "1"}, {"BWTC_HIP_FIN_PASSES": "0", "BWTC_HIP_TEXT_ROUNDS": "0"},
                {"BWTC_HIP_KEYS": "grams", "BWTC_HIP_LONG_G2": "1"}, {"BWTC_HIP_FINISHER": "0"}, {"BWTC_HIP_TEXT_ROUNDS": "2", "BWTC_HIP_FIN_PASSES": "2"},
                {"BWTC_HIP_KEYS": "grams"}, {"BWTC_HIP_CODE_BITS": "48", "BWTC_HIP_FIN_GROUP": "512"}, {"BWTC_HIP_FIN_FLOOR": "48", "BWTC_HIP_FIN_PASSES": "1"},
                {"BWTC_HIP_LOCAL_ROUNDS": "0"}, {"BWTC_HIP_FIN_WINDOW": "2048", "BWTC_HIP_FIN_GROUP": "1024", "BWTC_HIP_FIN_WORDS": "3"},
                {"BWTC_HIP_FIN_FLOOR": "0"}, {"BWTC_HIP_CODE_BITS": "64", "BWTC_HIP_TEXT_ROUNDS": "0"},
                {"BWTC_HIP_FIN_ROUNDS": "1"}, {"BWTC_HIP_FIN_ROUNDS": "2", "BWTC_HIP_FIN_PASSES": "1"},
                {"BWTC_HIP_FIN_WIDE": "0"}, {"BWTC_HIP_FIN_WIDE": "0", "BWTC_HIP_FIN_PASSES": "1"}]
    while "6" in phases and time.time() - t0 < budget:
        var = variants[n6 % len(variants)]
        os.environ["BWTC_HIP_GRAM_MIN_N"] = "64"
        for k in FUZZ6_KEYS:
            os.environ.pop(k, None)
        os.environ.update(var)
        with hip.Context(device=0, max_block_size=(4 << 20) + 64) as c6:
            for _ in range(12):
                n = int(rng.integers(200, int(rng.choice([3000, 60000, 900000, 4000000]))))
                sigma = int(rng.integers(5, 70)) if rng.random() < 0.7 else int(rng.integers(120, 240))
                alphabet = rng.choice(np.arange(0 if rng.random() < 0.3 else 1, 256), sigma, replace=False).astype(np.uint8)
                words = [alphabet[rng.integers(0, sigma, int(rng.integers(1, 12)))] for _ in range(int(rng.integers(2, 300)))]
                d = np.concatenate([words[int(i)] for i in rng.integers(0, len(words), n // 2 + 8)])[:n].copy()
                n = d.size
                for _ in range(int(rng.integers(0, 12))):
                    what = int(rng.integers(0, 5))
                    if what == 4:                              # an exotic but repetitive string: symbols from outside the alphabet, a short period, long
                        rare = np.setdiff1d(np.arange(1, 256), alphabet)[:int(rng.integers(1, 4))].astype(np.uint8)
                        if rare.size and n > 600:
                            a = int(rng.integers(0, n - 500))
                            ln = int(min(n - a, rng.integers(300, 20000)))
                            d[a:a + ln] = np.tile(rare, ln // rare.size + 1)[:ln]
                    elif what == 0:                            # a planted repeat
                        ln = int(min(n // 3, rng.integers(1, 1 + int(rng.choice([30, 1500, 200000])))))
                        a, b = int(rng.integers(0, n - ln)), int(rng.integers(0, n - ln))
                        d[b:b + ln] = d[a:a + ln].copy()
                    elif what == 1:                            # a periodic stretch
                        per = d[:int(rng.integers(1, 40))].copy()
                        a = int(rng.integers(0, n - 1))
                        ln = int(min(n - a, rng.integers(1, 1 + n // 2)))
                        d[a:a + ln] = np.tile(per, ln // per.size + 1)[:ln]
                    elif what == 2:                            # a run
                        a = int(rng.integers(0, n - 1))
                        d[a:a + int(rng.integers(1, 9000))] = alphabet[int(rng.integers(0, sigma))]
                    else:                                      # the block ends in its own beginning
                        ln = int(rng.integers(1, n // 2 + 1))
                        d[n - ln:] = d[:ln].copy()
                sp = int(rng.choice([1, 3, 8, 256]))
                try:
                    a = c6.bwt_block(d, sp)
                except hip.BwtcHipError as ex:
                    bad += 1
                    print("ERROR phase 6:", ex, "variant", var, "size", n, "sigma", sigma, "sp", sp, flush=True)
                    np.save(os.path.join(ROOT, "gpurun_out", "fuzz6_error_%d.npy" % n6), d)
                    n6 += 1
                    continue
                b = oracle_lib.oracle_bwt_block(d, sp)
                if not ((a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all()):
                    bad += 1
                    print("MISMATCH phase 6: variant", var, "size", n, "sigma", sigma, "sp", sp, "route", c6.stats().route, flush=True)
                    np.save(os.path.join(ROOT, "gpurun_out", "fuzz6_fail_%d.npy" % n6), d)
                if n < 600000:
                    opts = "p" * int(rng.integers(1, 5))
                    g, og = hip.Grammar(), oracle_lib.OracleGrammar()
                    mine = c6.precompress(g, opts, d)
                    want = oracle_lib.oracle_precompress(og, opts, d)
                    if mine.tobytes() != want.tobytes() or g.write().tobytes() != og.write().tobytes():
                        bad += 1
                        print("MISMATCH phase 6 prepr: size", n, "sigma", sigma, "options", opts, flush=True)
                        np.save(os.path.join(ROOT, "gpurun_out", "fuzz6_prepr_fail_%d.npy" % n6), d)
                n6 += 1
    for k in ("BWTC_HIP_GRAM_MIN_N",) + FUZZ6_KEYS:
        os.environ.pop(k, None)
    print("phase 6: %d blocks on the long-key route (and through the pre-stage)" % n6, flush=True)
    print("mismatches:", bad, flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
