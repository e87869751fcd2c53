"""Times the 'B' route (BWT + wavelet coder) on one block; BWTC_HIP_DEBUG=1 prints the phases."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from bwtc_amd import hip, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mib", type=int, default=256)
ap.add_argument("--kind", default="text")
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
n = a.mib << 20
data = {"text": synth.gen_text, "dna": synth.gen_dna, "random": synth.gen_random_bytes}[a.kind](n, 3)
ctx = hip.Context(device=0, max_block_size=n + 1024)
for i in range(a.reps):
    ctx.wavelet_reset()
    t = time.time()
    rec, _ = ctx.transform_and_encode_wavelet(data, 8, threads=a.threads) if a.threads else ctx.transform_and_encode_wavelet(data, 8)
    dt = time.time() - t
    print("rep %d: %.1f ms  %.1f MB/s  record %d bytes" % (i, dt * 1e3, n / 1e6 / dt, rec.size), flush=True)
ctx.close()
