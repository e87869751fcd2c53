"""Scratch: the overlapped 'B' route without torch in the process (does the runtime pick another
copy path for the big device-to-host copies when torch has initialised HIP first?)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
if os.environ.get("WITH_TORCH"):
    import torch  # noqa: E402
    torch.cuda.set_device(0)
    torch.empty(16, device="cuda")
from bwtc_amd import hip  # noqa: E402

n = 256 << 20
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
depth = int(os.environ.get("PROBE_DEPTH", "12"))
os.environ.setdefault("BWTC_HIP_WAVELET_DEPTH", str(depth))
ctx = hip.Context(0, n)
blocks = [hip.synth_into("t", 3 + j, ctx.host_alloc(n)) for j in range(2)]
d_in = [ctx.dmalloc(n + 64), ctx.dmalloc(n + 64)]
d_out = ctx.dmalloc(n + 64)
ring = [np.empty(ctx.compress_bound(n), np.uint8) for _ in range(depth + 1)]
pend = []
ctx.to_device_async(d_in[0], blocks[0])
marks = []
for i in range(steps):
    if len(pend) >= depth:
        ctx.wavelet_encode_end(pend.pop(0))
    ctx.copy_wait()
    ctx.to_device_async(d_in[(i + 1) % 2], blocks[(i + 1) % 2])
    lf, fr = ctx.bwt_block_device(d_in[i % 2], d_out, n, 8)
    pend.append(ctx.wavelet_encode_device_begin(d_out, n, lf, fr, ring[i % (depth + 1)], 16))
    marks.append(time.perf_counter())
while pend:
    ctx.wavelet_encode_end(pend.pop(0))
d = np.diff(marks)
print("ms per step (median of the last %d): %.2f" % (len(d) // 2, 1e3 * np.median(d[len(d) // 2:])))
