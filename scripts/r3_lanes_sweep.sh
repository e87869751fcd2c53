#!/bin/bash
# bench.py with the W lane engines holding one vector of chains (16) or two stepped alternately (32)
mkdir -p gpurun_out/r3_lanes
for cfg in "16 6" "32 6" "32 4" "32 3"; do
  set -- $cfg
  BWTC_HIP_W_LANES=$1 BWTC_HIP_W_ENGINES=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 64 > gpurun_out/r3_lanes/l$1_e$2.json 2> gpurun_out/r3_lanes/l$1_e$2.err || exit 1
  python - "$1" "$2" <<'P'
import json, sys
d = json.load(open("gpurun_out/r3_lanes/l%s_e%s.json" % (sys.argv[1], sys.argv[2])))
print("lanes", sys.argv[1], "engines", sys.argv[2], {k: d[k] for k in ("value", "ms_per_step", "gpu_ms_per_step", "host_core_s_per_block", "block_latency_ms", "drain_ms", "collect_wait_ms_per_step", "single_block_ms")})
P
done
