#!/bin/bash
# The driver's command (--steps 20 --warmup 5) at 16 / 20 / 24 blocks under way, three times each
mkdir -p gpurun_out/r3_depth
for rep in 1 2 3; do
for d in 16 20 24; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --depth $d > gpurun_out/r3_depth/d${d}_$rep.json 2> gpurun_out/r3_depth/d${d}_$rep.err || exit 1
  python - "$d" "$rep" <<'P'
import json, sys
d = json.load(open("gpurun_out/r3_depth/d%s_%s.json" % (sys.argv[1], sys.argv[2])))
print("depth", sys.argv[1], {k: d[k] for k in ("value", "ms_per_step", "gpu_ms_per_step", "host_core_s_per_block", "block_latency_ms", "collect_wait_ms_per_step", "host_blocks_finished_in_region", "host_staging_peak_gb_rank0")})
P
done
done
