#!/usr/bin/env python3
"""`compress` over a long stream from a pipe (the 4.3 GB synthetic file, `reps` times over), output thrown
away: wall time and the compressor's resident memory while it runs.  usage: r3_cli_soak.py [reps] [extra args]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bwtc_amd import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
extra = sys.argv[2:]
src = "/tmp/cli_soak_in.bin"
if not os.path.exists(src):
    with open(src, "wb") as f:
        for i in range(4):
            f.write(synth.gen_text(268435456, 100 + i).tobytes())
        f.flush()
for _ in range(3):                                   # into the page cache
    open(src, "rb").read(1 << 20)
exe = os.path.join(ROOT, "bwtc_amd", "host", "compress")
t0 = time.time()
cat = subprocess.Popen(["cat"] + [src] * reps, stdout=subprocess.PIPE)
comp = subprocess.Popen([exe, "-v", "1", "-m", "1451", "-e", "B", "-i", "-c"] + extra, stdin=cat.stdout, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
cat.stdout.close()
peak, samples = 0.0, []
while comp.poll() is None:
    try:
        for l in open("/proc/%d/status" % comp.pid):
            if l.startswith("VmRSS"):
                r = int(l.split()[1]) / 1e6
                peak = max(peak, r)
                samples.append(round(r, 1))
    except Exception:
        pass
    time.sleep(1.0)
err = comp.stderr.read().decode()
wall = time.time() - t0
gb = reps * 4 * 268435456 / 1e9
print("%s: %d blocks, %.1f GB in %.1f s = %.2f GB/s; resident memory each second (GB): %s; rc %d %s" %
      (" ".join(extra) or "no --devices (the CLI picks: farm with one context for a pipe)", reps * 4, gb, wall, gb / wall, samples, comp.returncode, err.strip().splitlines()[-1] if err.strip() else ""))
