#!/bin/bash
# default bench three times with the debug split (scalar / lanes)
mkdir -p gpurun_out/r3_ab
for rep in 1 2 3; do
  BWTC_HIP_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 64 > gpurun_out/r3_ab/c$rep.json 2> gpurun_out/r3_ab/c$rep.err || exit 1
  python - "$rep" <<'P'
import json, sys, re
d = json.load(open("gpurun_out/r3_ab/c%s.json" % sys.argv[1]))
err = open("gpurun_out/r3_ab/c%s.err" % sys.argv[1]).read()
m = re.findall(r"(\d+) blocks; host time in models [\d.]+ s, in range coders ([\d.]+) s", err)
s = re.findall(r"([\d.]+) s of the range coders' time in the scalar", err)
nb, tot, sc = int(m[-1][0]), float(m[-1][1]), float(s[-1])
print({k: d[k] for k in ("value", "ms_per_step", "gpu_ms_per_step", "host_core_s_per_block", "collect_wait_ms_per_step", "single_block_ms")}, "scalar %.3f lanes %.3f" % (sc / nb, (tot - sc) / nb))
P
done
