#!/usr/bin/env python3
"""Round 5: the transform's rate on real inputs -- one JSON line per workload: rounds, R_eff, route, device ms.
usage: workloads.py [MiB] [kind ...]      (BWTC_HIP_DEBUG=1 prints the sorter's own trace)
kinds: those of scripts/r4/workloads.py, plus
  pycorpus   the largest real text of the image: every .py file under the Python library directories
             (302 MB: standard library, torch, transformers, ...), in sorted walk order, cut at the block size"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bwtc_amd import hip  # noqa: E402
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("workloads_r4", os.path.join(ROOT, "scripts", "r4", "workloads.py"))
r4 = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(r4)

PY_ROOTS = ("/usr/lib/python3", "/usr/lib/python3.10", "/usr/local/lib/python3.10")


def py_corpus(limit):
    """Every .py file of the image's Python library directories, each real file once, sorted walk order."""
    path = os.environ.get("BWTC_CORPUS")
    if path and os.path.exists(path):
        with open(path, "rb") as f:
            return np.frombuffer(f.read(limit), np.uint8), "BWTC_CORPUS=%s" % path
    buf = bytearray()
    seen = set()
    files = 0
    for root in PY_ROOTS:
        for dirpath, dirnames, filenames in os.walk(root):
            dirnames[:] = sorted(d for d in dirnames if d != "__pycache__")
            for name in sorted(filenames):
                if len(buf) >= limit:
                    break
                if not name.endswith(".py"):
                    continue
                p = os.path.join(dirpath, name)
                rp = os.path.realpath(p)
                if rp in seen:
                    continue
                seen.add(rp)
                try:
                    with open(p, "rb") as f:
                        buf += f.read(limit - len(buf))
                        files += 1
                except OSError:
                    pass
    return np.frombuffer(bytes(buf[:limit]), np.uint8), "%d .py files of the image's Python libraries (%s), sorted walk order" % (files, ", ".join(PY_ROOTS))


def gen(kind, n):
    if kind == "pycorpus":
        d, what = py_corpus(n)
        if d.size < n:
            d = np.tile(d, n // d.size + 1)[:n]
            what += " (repeated to size)"
        return d.copy(), what
    if kind.startswith("c3copies"):
        # the generator's text with COPIES copies (default 300) of one 10 KB piece of it at random places: a short finisher
        # list (a few per cent of the block) whose groups are large AND deep -- the wide shape's worst case
        copies = int(kind[8:] or 300)
        d, _ = r4.gen("c3", n)
        d = d.copy()
        rng = np.random.default_rng(11)
        piece = d[12345:12345 + 10000].copy()
        for at in rng.integers(0, n - 10000, copies):
            d[at:at + 10000] = piece
        return d, "C3 generator with %d copies of a 10 KB piece at random places" % copies
    return r4.gen(kind, n)


def main():
    args = sys.argv[1:]
    mib = int(args[0]) if args and args[0].isdigit() else 256
    kinds = [a for a in args if not a.isdigit()] or ["c3", "realtext", "pycorpus"]
    n = mib << 20
    reps = int(os.environ.get("REPS", "3"))
    ctx = hip.Context(0, n)
    d_in, d_out = ctx.dmalloc(n + 64), ctx.dmalloc(n + 64)
    blk = ctx.host_alloc(n)
    for kind in kinds:
        data, what = gen(kind, n)
        blk[:] = data
        best = None
        for rep in range(reps):
            ctx.to_device_async(d_in, blk)
            ctx.copy_wait()
            lf, freqs = ctx.bwt_block_device(d_in, d_out, n, 8)
            st = ctx.stats()
            if best is None or st.ms_total < best.ms_total:
                best = hip.Stats.from_buffer_copy(bytes(st))
        verified = None
        if os.environ.get("VERIFY") == "1":
            # the GPU inverse (which checks every LF power against its own walk) gives the block back; freqs is its histogram
            ctx.inverse_bwt_block_device(d_out, d_in, n, lf)
            back = ctx.to_host(d_in, n)
            verified = bool((back == data).all() and (freqs == np.bincount(data, minlength=256)).all())
            if not verified:
                raise SystemExit("%s: the inverse transform does not give the block back" % kind)
        line = {"workload": kind, "what": what, "MiB": mib, "sigma": int(np.count_nonzero(np.bincount(data, minlength=256))),
                "device_ms_bwt": round(best.ms_total, 2),
                "MBps": round(n / 1e6 / (best.ms_total * 1e-3), 1), "rounds": best.rounds,
                "R_eff": round(best.active_sum / best.n, 3), "sort_passes_per_suffix": round(best.sort_pass_items / best.n, 1),
                "route": best.route, "alg_GB": round(best.alg_bytes / 1e9, 1),
                "alg_frac_of_8TBps": round(best.alg_bytes / (best.ms_total * 1e-3) / 8e12, 3)}
        if verified is not None:
            line["inverse_gives_the_block_back"] = verified
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
