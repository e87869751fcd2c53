"""How much CPU the GPU-feeding thread burns while it waits (BWTC_HIP_SYNC=spin|block)."""
import os
import sys
import time


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import numpy as np
    import torch
    from bwtc_amd import hip, synth
    n = 64 << 20
    dev = torch.device("cuda", 0)
    host = synth.gen_text(n, 3)
    d_in = torch.from_numpy(host).to(dev); d_out = torch.empty_like(d_in)
    ctx = hip.Context(0, n)
    ctx.bwt_block_device(d_in.data_ptr(), d_out.data_ptr(), n, 8)
    w0, c0 = time.perf_counter(), time.process_time()
    for _ in range(40):
        ctx.bwt_block_device(d_in.data_ptr(), d_out.data_ptr(), n, 8)
    w, c = time.perf_counter() - w0, time.process_time() - c0
    print("wall %.2f s, process cpu %.2f s -> %.0f%% of one CPU" % (w, c, 100 * c / w))


if __name__ == "__main__":
    main()
