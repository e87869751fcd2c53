#!/bin/bash
# The driver's command (--steps 20 --warmup 5) with 6 / 5 / 4 lane engines and 16 / 15 worker threads, twice each
mkdir -p gpurun_out/r3_eng
for rep in 1 2; do
for cfg in "6 16" "5 16" "4 16" "5 15"; do
  set -- $cfg
  BWTC_HIP_W_ENGINES=$1 BWTC_BENCH_THREADS=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r3_eng/e$1_t$2_$rep.json 2> gpurun_out/r3_eng/e$1_t$2_$rep.err || exit 1
  python - "$1" "$2" "$rep" <<'P'
import json, sys
d = json.load(open("gpurun_out/r3_eng/e%s_t%s_%s.json" % (sys.argv[1], sys.argv[2], sys.argv[3])))
print("engines", sys.argv[1], "threads", sys.argv[2], {k: d[k] for k in ("value", "ms_per_step", "gpu_ms_per_step", "host_core_s_per_block", "block_latency_ms", "collect_wait_ms_per_step")})
P
done
done
