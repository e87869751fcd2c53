#!/bin/bash
# what the last finisher change (branch-free comparison, every round's characters in one read) moved: the workloads
# table, the two real-text bench lines and their kernel statistics -> gpurun_out/r05_refresh/
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
O=gpurun_out/r05_refresh
mkdir -p $O
VERIFY=1 REPS=3 timeout -k 10 400 python3 scripts/r5/workloads.py 256 c3 realtext pycorpus dna random period9 zeros reptext > $O/workloads_bwt_256MiB.jsonl 2> $O/workloads.err || { tail -5 $O/workloads.err; exit 1; }
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --bwt-only --no-cpu-baseline > $O/n1_bwt_only_256.json 2> $O/n1_bwt_only_256.err || exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --bwt-only --workload realtext --no-cpu-baseline > $O/n1_bwt_only_realtext.json 2> $O/n1_bwt_only_realtext.err || exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --bwt-only --workload pycorpus --no-cpu-baseline > $O/n1_bwt_only_pycorpus.json 2> $O/n1_bwt_only_pycorpus.err || exit 1
for k in realtext pycorpus; do bash scripts/r5/prof_kind.sh $k 256 3 || exit 1; done
cp gpurun_out/r05_prof/bwt_kernel_stats_realtext.csv $O/kernel_stats_bwt_only_realtext256.csv
cp gpurun_out/r05_prof/bwt_kernel_stats_pycorpus.csv $O/kernel_stats_bwt_only_pycorpus256.csv
python3 -c "
import json
for l in open('$O/workloads_bwt_256MiB.jsonl'):
    d = json.loads(l); print(d['workload'], d['device_ms_bwt'], d['rounds'], d.get('verified'))
for k in ('256', 'realtext', 'pycorpus'):
    d = json.load(open('$O/n1_bwt_only_%s.json' % k)); print(k, d['value'], d['ms_per_step'], d['device_ms_bwt'], d['roofline']['frac'])
"
echo done
