#!/bin/bash
# the round's final build on a fresh box: the whole GPU suite, smoke(), the driver's bench command, the 100-step 'B'
# line, the transform alone, and the C3 kernel statistics -> gpurun_out/r05_final2/
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
O=gpurun_out/r05_final2
mkdir -p $O
timeout -k 10 780 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1 || { tail -30 $O/gputest.log; exit 1; }
tail -2 $O/gputest.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/n1_driver_command.json 2> $O/n1_driver_command.err || { tail -20 $O/n1_driver_command.err; exit 1; }
timeout -k 10 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > $O/n1_steps100.json 2> $O/n1_steps100.err || exit 1
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --bwt-only --no-cpu-baseline > $O/n1_bwt_only_256.json 2> $O/n1_bwt_only_256.err || exit 1
bash scripts/r5/prof_kind.sh c3 256 6 || exit 1
cp gpurun_out/r05_prof/bwt_kernel_stats_c3.csv $O/kernel_stats_bwt_only_text256.csv
python3 -c "
import json
for k in ('driver_command', 'steps100', 'bwt_only_256'):
    d = json.load(open('$O/n1_%s.json' % k)); print(k, d['value'], d['ms_per_step'], d['gpu_ms_per_step'], d.get('device_ms_bwt'), d['roofline']['frac'], d['host_bound'])
"
echo done
