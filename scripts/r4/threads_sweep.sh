#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
for t in 16 15 14; do
  BWTC_BENCH_THREADS=$t timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('threads $t:', d['value'], 'MB/s', d['ms_per_step'], 'ms/step gpu', d['gpu_ms_per_step'], 'wait', d['collect_wait_ms_per_step'], 'cpu-s/step', d['process_cpu_s_per_step'], d['cgroup_throttled'], 'finished', d['host_blocks_finished_in_region'], 'depth', d['config']['blocks_under_way'])
" || exit 1
done
