"""Per-phase wall times of the overlapped 'B' route (device half in the calling thread)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from bwtc_amd import hip, synth  # noqa: E402

n = 256 << 20
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0)
host = synth.gen_text(n, 3)
d_in = torch.from_numpy(host).to(dev)
d_out = torch.empty_like(d_in)
ctx = hip.Context(0, n)
depth = int(os.environ.get("PROBE_DEPTH", "16"))
ring = [np.empty(ctx.compress_bound(n), np.uint8) for _ in range(depth)]
pend = []
t0 = time.perf_counter()
for i in range(steps):
    a = time.perf_counter()
    lf, fr = ctx.bwt_block_device(d_in.data_ptr(), d_out.data_ptr(), n, 8)
    b = time.perf_counter()
    w = 0.0
    if len(pend) >= depth:
        ctx.wavelet_encode_end(pend.pop(0))
        w = time.perf_counter() - b
    c = time.perf_counter()
    ctx.wavelet_reset()
    pend.append(ctx.wavelet_encode_device_begin(d_out.data_ptr(), n, lf, fr, ring[i % depth], int(os.environ.get("PROBE_THREADS", "0"))))
    d = time.perf_counter()
    print("step %2d: bwt %.1f ms (device %.1f), wait oldest %.1f ms, begin %.1f ms" % (i, 1e3 * (b - a), ctx.stats().ms_total, 1e3 * w, 1e3 * (d - c)), flush=True)
while pend:
    ctx.wavelet_encode_end(pend.pop(0))
el = time.perf_counter() - t0
print("%d blocks in %.2f s: %.1f MB/s" % (steps, el, steps * n / 1e6 / el))
ctx.close()
