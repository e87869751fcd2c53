#!/usr/bin/env python3
"""Resident memory of a long 'B' stream through one context, printed every 50 blocks (no torch).
usage: rss_probe.py [blocks] [MiB per block]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bwtc_amd import hip, synth

def rss_gb():
    return int(open("/proc/self/statm").read().split()[1]) * 4096 / 1e9


def hwm_gb():
    for l in open("/proc/self/status"):
        if l.startswith("VmHWM"):
            return int(l.split()[1]) * 1024 / 1e9

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 400
size = (int(sys.argv[2]) if len(sys.argv) > 2 else 64) << 20
depth = int(os.environ.get("PROBE_DEPTH", "8"))
os.environ.setdefault("BWTC_HIP_WAVELET_DEPTH", str(depth))
ctx = hip.Context(device=0, max_block_size=size + 1024)
pool = [hip.synth_into("t", 3 + j, ctx.host_alloc(size)) for j in range(4)]
d_in = [ctx.dmalloc(size + 64), ctx.dmalloc(size + 64)]
d_out = ctx.dmalloc(size + 64)
if os.environ.get("PROBE_TORCH") == "1":
    import torch
    torch.cuda.set_device(0)
bound = ctx.compress_bound(size)
ring = [np.empty(bound, np.uint8) for _ in range(depth + 1)]
pending = []
t0 = time.time()
for i in range(blocks):
    if len(pending) >= depth:
        ctx.wavelet_encode_end(pending.pop(0))
    if i == 0:
        ctx.to_device_async(d_in[0], pool[0])
    ctx.copy_wait()
    ctx.to_device_async(d_in[(i + 1) % 2], pool[(i + 1) % 4])
    lf, freqs = ctx.bwt_block_device(d_in[i % 2], d_out, size, 8)
    pending.append(ctx.wavelet_encode_device_begin(d_out, size, lf, freqs, ring[i % (depth + 1)], 16))
    if i % 50 == 49:
        print("block %d: RSS %.2f GB (high-water %.2f), staging now/peak %.2f / %.2f GB, %.1f s" % ((i + 1, rss_gb(), hwm_gb()) + tuple(x / 1e9 for x in hip.host_staging_bytes()) + (time.time() - t0,)), flush=True)
while pending:
    ctx.wavelet_encode_end(pending.pop(0))
print("end: RSS %.2f GB" % rss_gb(), flush=True)
ctx.close()
print("closed: RSS %.2f GB" % rss_gb(), flush=True)
