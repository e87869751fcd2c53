#!/usr/bin/env python3
"""Peak resident memory of bench.py against the length of the stream (a leak would grow with it)."""
import json, resource, subprocess, sys, time
for steps in sys.argv[1:]:
    t = time.time()
    before = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    r = subprocess.run(["python", "bench.py", "--steps", steps, "--no-cpu-baseline"], capture_output=True, text=True)
    u = resource.getrusage(resource.RUSAGE_CHILDREN)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print("steps", steps, "rc", r.returncode, "wall %.1f s" % (time.time() - t), "max RSS so far %.1f GB" % (u.ru_maxrss / 1e6),
          "value", d["value"], "staging peak", d["host_staging_peak_gb_rank0"], flush=True)
