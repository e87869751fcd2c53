"""Round-4 scratch: does the context's blocking wait sleep?  CPU time of the calling thread against wall time
over ten 256 MiB transforms, BWTC_HIP_SYNC=spin and =block."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    from bwtc_amd import hip
    n = 256 << 20
    ctx = hip.Context(0, n)
    blk = hip.synth_into("t", 3, ctx.host_alloc(n))
    d_in, d_out = ctx.dmalloc(n + 64), ctx.dmalloc(n + 64)
    ctx.to_device_async(d_in, blk); ctx.copy_wait()
    ctx.bwt_block_device(d_in, d_out, n, 8)
    c0, w0 = time.thread_time(), time.perf_counter()
    for i in range(10):
        ctx.to_device_async(d_in, blk); ctx.copy_wait()
        ctx.bwt_block_device(d_in, d_out, n, 8)
    print(sys.argv[1], "thread CPU %.3f s of %.3f s wall; process CPU %.3f" % (time.thread_time() - c0, time.perf_counter() - w0, time.process_time()))
else:
    print({k: v for k, v in os.environ.items() if k.startswith(("HSA", "HIP", "ROC", "AMD", "GPU"))})
    for mode in ("spin", "block"):
        env = dict(os.environ, BWTC_HIP_SYNC=mode)
        subprocess.run([sys.executable, __file__, mode], env=env, check=False)
        for extra in ({"HSA_ENABLE_INTERRUPT": "1"},):
            if mode == "block":
                subprocess.run([sys.executable, __file__, mode + "+" + str(extra)], env=dict(env, **extra), check=False)
