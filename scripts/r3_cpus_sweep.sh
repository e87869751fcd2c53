#!/bin/bash
# Where the 16 worker threads run: the whole NUMA slice (default), sixteen first hardware threads, ...
mkdir -p gpurun_out/r3_cpus
lscpu | grep -E "Thread|Core|Socket|NUMA node[0-9]" > gpurun_out/r3_cpus/lscpu.txt
cat /sys/devices/system/cpu/cpu64/topology/thread_siblings_list >> gpurun_out/r3_cpus/lscpu.txt
for cfg in "default" "64-81" "64-95" "64-79,192-207"; do
  if [ "$cfg" = default ]; then unset BWTC_BENCH_CPULIST; else export BWTC_BENCH_CPULIST=$cfg; fi
  BWTC_HIP_DEBUG=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 64 > gpurun_out/r3_cpus/c.json 2> gpurun_out/r3_cpus/c.err || exit 1
  python - "$cfg" <<'P'
import json, sys, re
d = json.load(open("gpurun_out/r3_cpus/c.json"))
err = open("gpurun_out/r3_cpus/c.err").read()
m = re.findall(r"(\d+) blocks; host time in models [\d.]+ s, in range coders ([\d.]+) s", err)
s = re.findall(r"([\d.]+) s of the range coders' time in the scalar", err)
extra = ""
if m and s:
    nb, tot, sc = int(m[-1][0]), float(m[-1][1]), float(s[-1])
    extra = "scalar %.3f lanes %.3f core-s per block" % (sc / nb, (tot - sc) / nb)
print("cpus", sys.argv[1], {k: d[k] for k in ("value", "ms_per_step", "gpu_ms_per_step", "host_core_s_per_block", "collect_wait_ms_per_step")}, extra)
P
done
cat gpurun_out/r3_cpus/lscpu.txt
