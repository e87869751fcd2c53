#!/bin/bash
# the final build's remaining lines: 'H' coder, the 1 GiB block, the 'B' route's kernel statistics, and a campaign over
# every phase -> gpurun_out/r05_last/
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
O=gpurun_out/r05_last
mkdir -p $O
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --coder H --no-cpu-baseline > $O/n1_coderH.json 2> $O/n1_coderH.err || exit 1
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --bwt-only --size-mib 1024 --no-cpu-baseline > $O/n1_c5_1GiB_bwt_only.json 2> $O/n1_c5.err || exit 1
python3 -c "
import json
for k in ('coderH', 'c5_1GiB_bwt_only'):
    d = json.load(open('$O/n1_%s.json' % k)); print(k, d['value'], d['ms_per_step'], d['gpu_ms_per_step'], d.get('device_ms_bwt'), d['roofline']['frac'], d['host_bound'])
"
timeout -k 10 200 python scripts/fuzz_gpu_parity.py 20 70707 123456 > $O/fuzz.log 2>&1 || { tail -20 $O/fuzz.log; exit 1; }
grep -E "^seed|^phase|^mismatches" $O/fuzz.log
( cd /tmp && export TMPDIR=/tmp && PROBE_DEPTH=6 timeout -k 10 300 rocprofv3 --kernel-trace -d "$ROOT/$O/stats" -o s -- python3 "$ROOT/scripts/dev/pipe_notorch.py" 12 > "$ROOT/$O/stats.log" 2>&1 ) || { tail -5 $O/stats.log; exit 1; }
python3 scripts/rocpd_stats.py $O/stats/s_results.db $O/kernel_stats_B_default.csv || exit 1
rm -rf $O/stats
echo done
