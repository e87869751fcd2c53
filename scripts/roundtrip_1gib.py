"""One-off stress: a single 1 GiB text block through compress (default coder 'B') and uncompress."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from bwtc_amd import synth  # noqa: E402

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = sys.argv[1] if len(sys.argv) > 1 else "/tmp"
coder = sys.argv[2] if len(sys.argv) > 2 else "B"
data = synth.gen_text(1 << 30, 5)
src, dst, out = (os.path.join(tmp, n) for n in ("t1g.bin", "t1g.bwtc", "t1g.out"))
data.tofile(src)
t = time.time()
r = subprocess.run([os.path.join(root, "bwtc_amd/host/compress"), "-m", "5806", "-e", coder, "-v", "1", src, dst],
                   capture_output=True, text=True)
print("compress rc", r.returncode, r.stderr.strip()[-300:], "%.1f s" % (time.time() - t), flush=True)
assert r.returncode == 0
print("compressed bytes", os.path.getsize(dst), flush=True)
t = time.time()
r = subprocess.run([os.path.join(root, "bwtc_amd/host/uncompress"), "-v", "1", dst, out], capture_output=True, text=True)
print("uncompress rc", r.returncode, r.stderr.strip()[-300:], "%.1f s" % (time.time() - t), flush=True)
assert r.returncode == 0
back = np.fromfile(out, np.uint8)
print("round trip", "OK" if back.size == data.size and (back == data).all() else "MISMATCH")
for f in (src, dst, out):
    os.remove(f)
