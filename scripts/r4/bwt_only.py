"""Round-4 scratch: N transforms of one synthetic block, device to device (for rocprofv3 kernel
statistics of the transform alone).  usage: bwt_only.py [steps] [MiB] [kind t|d|r]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bwtc_amd import hip  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = (int(sys.argv[2]) if len(sys.argv) > 2 else 256) << 20
kind = sys.argv[3] if len(sys.argv) > 3 else "t"
ctx = hip.Context(0, n)
blk = hip.synth_into(kind, 3, ctx.host_alloc(n))
d_in = ctx.dmalloc(n + 64)
d_out = ctx.dmalloc(n + 64)
ms = []
for i in range(steps):
    ctx.to_device_async(d_in, blk)
    ctx.copy_wait()
    ctx.bwt_block_device(d_in, d_out, n, 8)
    ms.append(ctx.stats().ms_total)
print("device ms per transform:", " ".join("%.2f" % x for x in ms))
