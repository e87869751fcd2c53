#!/bin/bash
# host half: lane engines x lanes per engine, paired long chains -- with the process's real CPU time per step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
run() {
  env "$@" timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline ${DEPTH:+--depth $DEPTH} 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$*', 'depth', d['config']['blocks_under_way'], ':', d['value'], 'MB/s', d['ms_per_step'], 'ms/step gpu', d['gpu_ms_per_step'], 'wait', d['collect_wait_ms_per_step'], 'core-s', d['host_core_s_per_block'], 'cpu-s/step', d['process_cpu_s_per_step'], d['cgroup_throttled'], 'finished', d['host_blocks_finished_in_region'], 'lat', d['block_latency_ms'])
" || exit 1
}
run BWTC_HIP_W_ENGINES=4
run BWTC_HIP_W_ENGINES=2 BWTC_HIP_W_LANES=32
run BWTC_HIP_W_ENGINES=3 BWTC_HIP_W_LANES=32
DEPTH=28 run BWTC_HIP_W_ENGINES=4 BWTC_HIP_W_PAIR_ENGINES=6
DEPTH=28 run BWTC_HIP_W_ENGINES=4 BWTC_HIP_W_PAIR_ENGINES=8
DEPTH=28 run BWTC_HIP_W_ENGINES=4
