#!/usr/bin/env python3
"""Round 4: the transform's rate on inputs harder (and easier) than the 64-token generator -- one JSON line per
workload: rounds, R_eff, route, device ms of the transform, MB/s.  usage: workloads.py [MiB] [kind ...]
kinds: c3 realtext realtext_rep dna random zeros period9 reptext"""
import json
import os
import sys
import sysconfig

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bwtc_amd import hip, synth  # noqa: E402


def real_text(limit):
    """Text written by people: BWTC_CORPUS when given, else the Python standard library's sources and the licence
    texts on the box, in sorted order."""
    path = os.environ.get("BWTC_CORPUS")
    if path and os.path.exists(path):
        with open(path, "rb") as f:
            return np.frombuffer(f.read(limit), np.uint8)
    buf = bytearray()
    for root in (sysconfig.get_paths()["stdlib"], "/usr/share/common-licenses"):
        for dirpath, dirnames, filenames in os.walk(root):
            dirnames[:] = sorted(d for d in dirnames if d not in ("site-packages", "dist-packages", "__pycache__"))
            for name in sorted(filenames):
                if len(buf) >= limit:
                    break
                if name.endswith((".py", ".txt", ".rst")) or root.endswith("licenses"):
                    try:
                        with open(os.path.join(dirpath, name), "rb") as f:
                            buf += f.read(limit - len(buf))
                    except OSError:
                        pass
    return np.frombuffer(bytes(buf), np.uint8)


def gen(kind, n):
    if kind == "c3":
        return synth.gen_text(n, 3), "C3 generator (64 tokens)"
    if kind == "dna":
        return synth.gen_dna(n, 2), "C2 generator (uniform ACGT)"
    if kind == "random":
        return synth.gen_random_bytes(n, 1), "C1 generator (random bytes)"
    if kind == "zeros":
        return np.zeros(n, np.uint8), "all-equal block"
    if kind == "period9":
        return np.tile(np.frombuffer(b"abcabcabd", np.uint8), n // 9 + 1)[:n].copy(), "period 9"
    if kind == "reptext":
        base = synth.gen_text(1 << 20, 3)
        return np.tile(base, n // base.size + 1)[:n].copy(), "1 MiB of C3 text repeated (test/CompressorAndDecompressorTest.cpp:52-59 in spirit)"
    base = real_text(n)
    if kind == "realtext":
        # the sources once, then again with their lines in another order each time: every line occurs several times
        # (common prefixes of a line's length), no paragraph does
        rng = np.random.default_rng(1)
        lines = bytes(base).split(b"\n")
        parts, total = [bytes(base)], base.size
        while total < n:
            perm = rng.permutation(len(lines))
            chunk = b"\n".join(lines[i] for i in perm)
            parts.append(chunk)
            total += len(chunk)
        return np.frombuffer(b"".join(parts)[:n], np.uint8).copy(), "real text (%d MB of Python sources and licences), then its lines shuffled, to size" % (base.size // 1000000)
    if kind == "realtext_rep":
        return np.tile(base, n // base.size + 1)[:n].copy(), "real text (%d MB) repeated to size" % (base.size // 1000000)
    raise SystemExit("unknown kind " + kind)


def main():
    args = sys.argv[1:]
    mib = int(args[0]) if args and args[0].isdigit() else 256
    kinds = [a for a in args if not a.isdigit()] or ["c3", "realtext", "dna", "random", "period9", "zeros", "reptext"]
    n = mib << 20
    ctx = hip.Context(0, n)
    d_in, d_out = ctx.dmalloc(n + 64), ctx.dmalloc(n + 64)
    for kind in kinds:
        data, what = gen(kind, n)
        blk = ctx.host_alloc(n)
        blk[:] = data
        best = None
        for rep in range(3):
            ctx.to_device_async(d_in, blk)
            ctx.copy_wait()
            ctx.bwt_block_device(d_in, d_out, n, 8)
            st = ctx.stats()
            if best is None or st.ms_total < best.ms_total:
                best = hip.Stats.from_buffer_copy(bytes(st))
        line = {"workload": kind, "what": what, "MiB": mib, "device_ms_bwt": round(best.ms_total, 2),
                "MBps": round(n / 1e6 / (best.ms_total * 1e-3), 1), "rounds": best.rounds,
                "R_eff": round(best.active_sum / best.n, 3), "sort_passes_per_suffix": round(best.sort_pass_items / best.n, 1),
                "route": best.route, "alg_GB": round(best.alg_bytes / 1e9, 1),
                "alg_frac_of_8TBps": round(best.alg_bytes / (best.ms_total * 1e-3) / 8e12, 3),
                "long_keys": os.environ.get("BWTC_HIP_LONG", "1") != "0"}
        print(json.dumps(line), flush=True)
        ctx.host_free(blk) if hasattr(ctx, "host_free") else None


if __name__ == "__main__":
    main()
