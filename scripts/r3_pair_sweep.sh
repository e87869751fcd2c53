#!/bin/bash
# bench.py with the long sections' scalar chains alone (0) or paired (n engines), at two depths
mkdir -p gpurun_out/r3_pair
for cfg in "0 16" "6 16" "6 24" "8 24" "0 24"; do
  set -- $cfg
  BWTC_HIP_W_PAIR_ENGINES=$1 timeout -k 10 200 python bench.py --no-cpu-baseline --depth $2 --steps 64 > gpurun_out/r3_pair/p$1_d$2.json 2> gpurun_out/r3_pair/p$1_d$2.err || exit 1
  python - "$1" "$2" <<'P'
import json, sys
d = json.load(open("gpurun_out/r3_pair/p%s_d%s.json" % (sys.argv[1], sys.argv[2])))
print("pair engines", sys.argv[1], "depth", sys.argv[2], {k: d[k] for k in ("value", "ms_per_step", "gpu_ms_per_step", "host_core_s_per_block", "block_latency_ms", "drain_ms", "collect_wait_ms_per_step", "single_block_ms")})
P
done
