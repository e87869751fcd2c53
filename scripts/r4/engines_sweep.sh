#!/bin/bash
# lane engines / scalar tasks of the host half against the shorter GPU side of round 4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
for e in 6 4 3 2; do
  BWTC_HIP_W_ENGINES=$e timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('engines $e:', d['value'], 'MB/s', d['ms_per_step'], 'ms/step gpu', d['gpu_ms_per_step'], 'wait', d['collect_wait_ms_per_step'], 'core-s', d['host_core_s_per_block'], 'cpu-s/step', d['process_cpu_s_per_step'], d['cgroup_throttled'], 'finished', d['host_blocks_finished_in_region'], 'lat', d['block_latency_ms'])
" || exit 1
done
